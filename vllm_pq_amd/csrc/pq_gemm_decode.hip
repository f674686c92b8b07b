// Host side of the decode kernel (pq_gemm_decode_impl.cuh): which problems it takes, its launch geometry.
#include <atomic>

#include "pq_gemm_decode_impl.cuh"
#include "pq_knobs.h"

namespace pq {

bool gemm_decode_supported(const GemmArgs &a) {
    if (!a.packed || a.d != 8 || a.M != 1 || a.code_bytes != 1) return false;
    if (a.K < 16 || a.K > 256 || (a.K & (a.K - 1))) return false;
    if (a.T < 1 || a.T > DC_MAX_TOKENS || a.G < 64 || a.G % 64) return false;          // whole 512-feature chunks
    if (a.x_stride % 8 || (uint64_t)a.x_stride * 2 * 16 >= (1ull << 31)) return false;
    if (a.N <= 0 || (int64_t)a.N * 2 >= (1ll << 31)) return false;
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.codes) | reinterpret_cast<uintptr_t>(a.codebooks)) & 15)
        return false;
    if ((reinterpret_cast<uintptr_t>(a.scales) | reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.y)) & 7) return false;
    if (a.y_stride % 4) return false;
    for (int p = 0; p < a.parts.num; ++p)
        if (a.parts.size[p] % 16) return false;
    if (a.G / 64 <= DC_MAX_WAVES) return true;   // one chunk per wave: K <= 4096
    return choose_decode_geom(a).nf > 0;         // longer K: split over blocks, which needs the workspace
}

// Measured against the stream kernel on the same launches (tools/decode_ab.py, profiles/r02_decode_ab.csv): ahead by
// 0.5 - 2 us (up to 30 %) at 1 .. 16 tokens wherever a block has at least three chunks or the layer more than 4096 rows,
// K split over blocks included; two-chunk layers of few rows are level and stay with the stream kernel.
bool gemm_decode_preferred(const GemmArgs &a) {
    if (!gemm_decode_supported(a)) return false;
    if (a.T > 16) return false;                  // token groups: chosen by the caller's comparison of modelled durations
    return a.G / 64 > 2 || a.N > 4096;
}

// Modelled duration in microseconds of the launch choose_decode_geom() picks, for the comparison with the other
// kernels at 17 .. DC_MAX_TOKENS tokens (every group of 16 tokens gathers the weights again, so this kernel is for
// layers whose blocks hold few units): measured 5.0 us for a one-unit block at 4096 inputs, 0.45 us per further unit
// of a wave, 2 us for the hand-over of a K split.
double decode_model_us(const GemmArgs &a) {
    const DecodeGeom g = choose_decode_geom(a);
    if (g.nf == 0) return 0.0;
    const long blocks = (long)tiles_in(a.parts, (g.waves >> g.wk_log2) * g.nf * 16) * g.ksplit * ((a.T + 15) / 16);
    const double rounds = (double)((blocks + 255) / 256);
    return rounds * (3.2 + 0.22 * g.cpb + 0.45 * g.nf) + (g.ksplit > 1 ? 2.0 : 0.0);
}

// Blocks per tile for K longer than a block's 8 chunks: the fewest that cover K with at most 8 chunks each and,
// beyond that, as many as still leave every tile's grid within one round of the chip (each block then reads a
// shorter slice of x); 1 without a workspace that holds the partial sums.
static int decode_ksplit(const GemmArgs &a, int nchunks) {
    if (nchunks <= DC_MAX_WAVES) return 1;
    const int need = (nchunks + DC_MAX_WAVES - 1) / DC_MAX_WAVES;
    if (need > 8) return 0;
    if (!a.workspace || reinterpret_cast<uintptr_t>(a.workspace) & 15) return 0;
    if (a.workspace_bytes < WS_COUNTER_BYTES + (size_t)need * a.T * a.N * 4) return 0;
    return need;
}

size_t decode_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes) {
    if (d != 8 || M != 1 || code_bytes != 1 || K < 16 || K > 256 || T < 1 || T > DC_MAX_TOKENS || G % 64) return 0;
    const int nchunks = G / 64;
    if (nchunks <= DC_MAX_WAVES || nchunks > 8 * DC_MAX_WAVES) return 0;
    const size_t part = (size_t)((nchunks + DC_MAX_WAVES - 1) / DC_MAX_WAVES) * T * N * 4;
    return part > WS_PARTIAL_BYTES ? 0 : WS_COUNTER_BYTES + part;
}

// Waves along K: as many as there are chunks, up to 8 (a power of two; three chunks take four waves).  Then the
// block shape with the shortest modelled launch: rounds of one block per CU, each a fixed cost plus its units in
// sequence.  nf <= WK (one output fragment per wave in the epilogue) and nf * chunks per wave <= DC_UMAX.
DecodeGeom choose_decode_geom(const GemmArgs &a) {
    const int nchunks = a.G / 64;
    DecodeGeom g;
    g.c = 1;
    g.ksplit = decode_ksplit(a, nchunks);
    if (g.ksplit == 0) { g.nf = 0; g.waves = 0; g.wk_log2 = 0; g.cpb = 0; return g; }    // not taken
    g.cpb = (nchunks + g.ksplit - 1) / g.ksplit;              // chunks per block
    g.wk_log2 = g.cpb >= 5 ? 3 : g.cpb >= 3 ? 2 : g.cpb >= 2 ? 1 : 0;
    const int WK = 1 << g.wk_log2;
    double best = -1.0;
    g.nf = 0;                     // not taken unless a shape passes the checks below
    g.waves = WK;
    const int tgroups = (a.T + 15) / 16;
    // a K-split launch has one arrival counter per tile and token group in the workspace's counter area
    auto counters_fit = [&](int wr, int nf) {
        return g.ksplit == 1 || (long)tiles_in(a.parts, wr * nf * 16) * tgroups * 4 <= (long)WS_COUNTER_BYTES;
    };
    if (knobs().decode_waves) {   // PQ_AMD_DECODE_CFG = "WAVESxNF", for tuning
        const int w = knobs().decode_waves, nf = knobs().decode_nf;
        if (w % WK == 0 && w >= WK && w <= DC_MAX_WAVES && !((w / WK) & (w / WK - 1)) && nf >= 1 && nf <= WK && nf * g.c <= DC_UMAX &&
            counters_fit(w / WK, nf)) {
            g.nf = nf;
            g.waves = w;
            return g;
        }
    }
    for (int wr = 1; wr * WK <= DC_MAX_WAVES; wr *= 2) {
        for (int nf = 1; nf <= WK && nf * g.c <= DC_UMAX; ++nf) {
            const long tiles = (long)tiles_in(a.parts, wr * nf * 16) * g.ksplit * tgroups;
            if (!counters_fit(wr, nf)) continue;
            const double rounds = (double)((tiles + 255) / 256);
            // fixed: launch ramp, image, epilogue; per unit of a wave about 0.3 us; blocks of few waves share a CU
            const double cost = rounds * (3.0 + 0.3 * nf * g.c) + (tiles < 256 ? 0.002 * (256 - tiles) : 0.0) * nf * g.c;
            if (best < 0.0 || cost < best) { best = cost; g.nf = nf; g.waves = wr * WK; }
        }
    }
    return g;
}

bool gemm_decode_norm_supported(const GemmArgs &a) {
    if (!a.norm || !gemm_decode_supported(a)) return false;
    const NormFuse &n = *a.norm;
    const DecodeGeom g = choose_decode_geom(a);
    if (g.nf == 0) return false;
    if (n.role == 1) {
        if (!n.residual || !n.norm_w || !n.xw || !n.ssq) return false;
        if ((reinterpret_cast<uintptr_t>(n.residual) | reinterpret_cast<uintptr_t>(n.norm_w) | reinterpret_cast<uintptr_t>(n.xw)) & 7) return false;
        if (n.res_stride % 4 || n.xw_stride % 4 || n.res_stride < a.N || n.xw_stride < a.N) return false;
        return (uint64_t)a.T * (uint64_t)n.res_stride * 2 < (1ull << 31);
    }
    if (n.role == 2) {
        // every thread adds at most DC_SSQ_LOADS fragments' sums for its token (8192 inputs with a block of 8 waves)
        return n.ssq && n.hidden == a.G * a.d && n.hidden % 16 == 0 && n.hidden / 16 <= DC_SSQ_LOADS * (g.waves * 4);
    }
    return false;
}

template <typename AT>
static hipError_t launch_decode_at(const GemmArgs &a, const DecodeGeom &g, hipStream_t stream, bool act = false) {
    DecodeParams p;
    p.act_rows = act ? a.parts.size[0] : 0;
    p.phases = 2;
    p.residual = nullptr; p.res_stride = 0; p.norm_w = nullptr; p.xw = nullptr; p.xw_stride = 0; p.ssq = nullptr;
    p.t_pad = (a.T + 15) & ~15; p.ssq_frags = 0; p.eps = 0.f; p.inv_hidden = 0.f;
    const int role = a.norm ? a.norm->role : 0;
    if (role == 1) {
        p.residual = static_cast<uint16_t *>(a.norm->residual); p.res_stride = a.norm->res_stride;
        p.norm_w = static_cast<const uint16_t *>(a.norm->norm_w);
        p.xw = static_cast<uint16_t *>(a.norm->xw); p.xw_stride = a.norm->xw_stride;
        p.ssq = a.norm->ssq;
    } else if (role == 2) {
        p.ssq = a.norm->ssq; p.ssq_frags = a.norm->hidden / 16; p.eps = a.norm->eps; p.inv_hidden = 1.0f / (float)a.norm->hidden;
    }
    p.x = static_cast<const unsigned char *>(a.x);
    p.codes = static_cast<const unsigned char *>(a.codes);
    p.codebooks = static_cast<const unsigned char *>(a.codebooks);
    p.scales = static_cast<const uint16_t *>(a.scales);
    p.bias = static_cast<const uint16_t *>(a.bias);
    p.y = static_cast<uint16_t *>(a.y);
    p.y_stride = a.y_stride;
    p.x_stride_bytes = (uint32_t)(a.x_stride * 2);
    p.T = a.T;
    p.nchunks = a.G / 64;
    p.K = a.K;
    p.nf = g.nf;
    p.wk_log2 = g.wk_log2;
    p.img_bytes = (uint32_t)a.K * 256u;
    p.spt = a.K * 16 >= g.waves * 64 ? a.K * 16 / (g.waves * 64) : 1;
    p.aux_off = (uint32_t)dc_lds_main_bytes(g.waves, p.img_bytes);
    p.ksplit = g.ksplit;
    p.cpb = g.cpb;
    p.n_total = a.N;
    p.ws_cnt = static_cast<unsigned *>(a.workspace);
    p.ws_part = g.ksplit > 1 ? reinterpret_cast<float *>(static_cast<unsigned char *>(a.workspace) + WS_COUNTER_BYTES) : nullptr;
    const int rows = (g.waves >> g.wk_log2) * g.nf * 16;
    Parts tparts = a.parts;
    if (act) {   // tiles over the gate partition; the up rows follow act_rows further on
        tparts.num = 1;
        for (int q = 1; q < PQ_MAX_PARTS; ++q) tparts.size[q] = 0;
    }
    p.tmap = make_tile_map(tparts, rows);
    const dim3 grid(tiles_in(tparts, rows), g.ksplit, (a.T + 15) / 16);
    const size_t lds = dc_lds_bytes(g.waves, p.img_bytes);
    // More than the default 64 KiB of LDS needs the opt-in once per kernel VARIANT and device: the variants share
    // one function-pointer type (and so one instantiation of this lambda), hence a flag array per variant index.
    auto launch = [&](auto kern, int variant) -> hipError_t {
        static std::atomic<bool> attr_set[5][64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[variant][dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)dc_lds_bytes(DC_MAX_WAVES, 65536));
            if (e != hipSuccess) return e;
            attr_set[variant][dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, grid, dim3(g.waves * 64), lds, stream, p);
        return hipGetLastError();
    };
    if (act) return role == 2 ? launch(gemm_decode_kernel<AT, false, 0, 2, true>, 4) : launch(gemm_decode_kernel<AT, false, 0, 0, true>, 3);
    if (role == 1) return launch(gemm_decode_kernel<AT, false, 0, 1>, 1);
    if (role == 2) return launch(gemm_decode_kernel<AT, false, 0, 2>, 2);
    return launch(gemm_decode_kernel<AT>, 0);
}

// Geometry of the fused gate_up + activation launch: 8 waves, as many along K as there are chunks, nf gate fragments
// (and as many up fragments) per wave with the fewest rounds of one block per CU.
static DecodeGeom choose_decode_act_geom(const GemmArgs &a) {
    DecodeGeom g{0, 0, 1, 0, 1, 0};
    const int nchunks = a.G / 64;
    if (nchunks < 1 || nchunks > DC_MAX_WAVES) return g;
    g.cpb = nchunks;
    g.wk_log2 = nchunks >= 5 ? 3 : nchunks >= 3 ? 2 : nchunks >= 2 ? 1 : 0;
    const int WK = 1 << g.wk_log2, wr = DC_MAX_WAVES / WK;
    g.waves = DC_MAX_WAVES;
    Parts gate;
    gate.num = 1;
    for (int q = 0; q < PQ_MAX_PARTS; ++q) gate.size[q] = q == 0 ? a.parts.size[0] : 0;
    double best = -1.0;
    for (int nf = 1; nf <= WK && 2 * nf <= DC_UMAX; ++nf) {
        const long tiles = (long)tiles_in(gate, wr * nf * 16) * ((a.T + 15) / 16);
        const double cost = (double)((tiles + 255) / 256) * (3.5 + 0.6 * nf) + (tiles < 256 ? 0.002 * (256 - tiles) : 0.0) * nf;
        if (best < 0.0 || cost < best) { best = cost; g.nf = nf; }
    }
    return g;
}

bool gemm_decode_act_supported(const GemmArgs &a) {
    if (a.parts.num != 2 || a.parts.size[0] != a.parts.size[1] || a.parts.size[0] % 16) return false;
    if (a.K != 256 || a.G / 64 > DC_MAX_WAVES) return false;
    GemmArgs full = a;                      // the plain kernel's format and alignment checks, on the [T][N] problem
    full.norm = nullptr;
    if (!gemm_decode_supported(full)) return false;
    if (a.norm) {
        if (a.norm->role != 2 || !a.norm->ssq || a.norm->hidden != a.G * a.d || a.norm->hidden % 16 ||
            a.norm->hidden / 16 > 8 * (DC_MAX_WAVES * 4))
            return false;
    }
    return choose_decode_act_geom(a).nf > 0;
}

hipError_t launch_gemm_decode_act(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    if (!gemm_decode_act_supported(a)) return hipErrorInvalidValue;
    const DecodeGeom g = choose_decode_act_geom(a);
    return a.dtype == PQ_DTYPE_BF16 ? launch_decode_at<bf16_t>(a, g, stream, true) : launch_decode_at<fp16_t>(a, g, stream, true);
}

hipError_t launch_gemm_decode(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    if (a.norm && !gemm_decode_norm_supported(a)) return hipErrorInvalidValue;
    const DecodeGeom g = choose_decode_geom(a);
    if (g.nf == 0) return hipErrorInvalidValue;
    return a.dtype == PQ_DTYPE_BF16 ? launch_decode_at<bf16_t>(a, g, stream) : launch_decode_at<fp16_t>(a, g, stream);
}

}  // namespace pq
