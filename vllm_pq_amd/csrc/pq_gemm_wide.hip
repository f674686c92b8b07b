// Host side of the mid-batch kernel (pq_gemm_wide_impl.cuh): which problems it takes, its launch geometry.
#include <atomic>

#include "pq_gemm_wide_impl.cuh"
#include "pq_knobs.h"

namespace pq {

bool gemm_wide_supported(const GemmArgs &a) {
    if (!a.packed || a.d != 8 || a.M != 1 || a.code_bytes != 1 || a.norm) return false;
    if (a.K < 16 || a.K > 256 || (a.K & (a.K - 1))) return false;
    if (a.T < 1 || a.G < 64 || a.G % 64 || a.G / 64 > WD_WAVES) return false;        // whole 512-feature chunks, one per wave
    if (a.x_stride % 8) return false;
    if (a.N <= 0 || (int64_t)a.N * 2 >= (1ll << 31)) return false;
    if ((int64_t)a.T * (a.N / 16 + 8) >= (1ll << 30)) return false;                   // blocks of the one-dimensional grid
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.codes) | reinterpret_cast<uintptr_t>(a.codebooks)) & 15)
        return false;
    if ((reinterpret_cast<uintptr_t>(a.scales) | reinterpret_cast<uintptr_t>(a.bias)) & 3) return false;
    if ((reinterpret_cast<uintptr_t>(a.y) & 7) || a.y_stride % 4) return false;
    for (int p = 0; p < a.parts.num; ++p)
        if (a.parts.size[p] % 16) return false;
    return true;
}

// Waves along K: as many as there are chunks (a power of two; three chunks take four waves), the other waves along
// the rows.  Then fragments per wave and round (nf) and rounds per block with the shortest modelled launch: the blocks
// run in rounds of one per CU; a block costs a fixed part (launch ramp, the two activation fragments, the image, the
// tail), and per round of units nf units in sequence plus the hand-over through LDS.
WideGeom choose_wide_geom(const GemmArgs &a, double *model_us) {
    WideGeom g{0, 0, 0};
    const int nchunks = a.G / 64;
    g.wk_log2 = nchunks >= 5 ? 3 : nchunks >= 3 ? 2 : nchunks >= 2 ? 1 : 0;
    const int WR = WD_WAVES >> g.wk_log2;
    const long tgroups = (a.T + WD_TOKENS - 1) / WD_TOKENS;
    if (knobs().wide_nf > 0) {   // PQ_AMD_WIDE_CFG = "NFxROUNDS", for tuning
        const int nf = knobs().wide_nf, rounds = knobs().wide_rounds;
        if (nf <= WD_MAX_NF && rounds >= 1 && WR * rounds * nf <= WD_MAX_FRAGS) {
            g.nf = nf;
            g.rounds = rounds;
            if (model_us) *model_us = 0.0;
            return g;
        }
    }
    double best = -1.0;
    for (int nf = 1; nf <= WD_MAX_NF; ++nf) {
        for (int rounds = 1; WR * rounds * nf <= WD_MAX_FRAGS; ++rounds) {
            const long blocks = (long)tiles_in(a.parts, WR * rounds * nf * 16) * tgroups;
            const double passes = (double)((blocks + 255) / 256);
            // measured on the MI355X (tools/wide_lab, launch to launch): about 7 us around the rounds (launch, the two
            // activation fragments, the image, the tail), 0.65 us per unit of a wave, 0.7 us per round's hand-over
            const double blk = 7.0 + rounds * (0.65 * nf + 0.7);
            const double cost = passes * blk;
            if (best < 0.0 || cost < best - 1e-9) { best = cost; g.nf = nf; g.rounds = rounds; }
        }
    }
    if (model_us) *model_us = best;
    return g;
}

template <typename AT, int NF>
static hipError_t launch_wide_nf(const GemmArgs &a, const WideGeom &g, hipStream_t stream) {
    WideParams p;
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
    p.rounds = g.rounds;
    p.wk_log2 = g.wk_log2;
    p.img_bytes = (uint32_t)a.K * 256u;
    p.spt = a.K * 16 >= WD_WAVES * 64 ? a.K * 16 / (WD_WAVES * 64) : 1;
    p.wave_bytes = (uint32_t)wd_wave_bytes(g.nf);
    p.ring = wd_ring(g.nf);
    const int frags_blk = (WD_WAVES >> g.wk_log2) * g.rounds * g.nf;
    p.sc_off = p.img_bytes + WD_WAVES * p.wave_bytes;
    const int rows = frags_blk * 16;
    p.tmap = make_tile_map(a.parts, rows);
    p.n_tiles = tiles_in(a.parts, rows);
    p.tgroups = (a.T + WD_TOKENS - 1) / WD_TOKENS;
    // whole groups of 8 row tiles (one per XCD), each with its token groups back to back on that XCD
    const dim3 grid((unsigned)((p.n_tiles + 7) / 8 * 8 * p.tgroups));
    const size_t lds = wd_lds_bytes(p.img_bytes, g.nf, frags_blk);
    auto kern = gemm_wide_kernel<AT, NF>;
    static std::atomic<bool> attr_set[64];   // per device, act dtype and NF (this function is a template): more than 64 KiB of LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)wd_lds_bytes(65536, WD_MAX_NF, WD_MAX_FRAGS));
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, dim3(WD_WAVES * 64), lds, stream, p);
    return hipGetLastError();
}

template <typename AT>
static hipError_t launch_wide_at(const GemmArgs &a, const WideGeom &g, hipStream_t stream) {
    switch (g.nf) {
    case 1: return launch_wide_nf<AT, 1>(a, g, stream);
    case 2: return launch_wide_nf<AT, 2>(a, g, stream);
    case 3: return launch_wide_nf<AT, 3>(a, g, stream);
    default: return launch_wide_nf<AT, 4>(a, g, stream);
    }
}

hipError_t launch_gemm_wide(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    if (!gemm_wide_supported(a)) return hipErrorInvalidValue;
    const WideGeom g = choose_wide_geom(a, nullptr);
    if (g.nf < 1 || g.nf > WD_MAX_NF) return hipErrorInvalidValue;
    return a.dtype == PQ_DTYPE_BF16 ? launch_wide_at<bf16_t>(a, g, stream) : launch_wide_at<fp16_t>(a, g, stream);
}

}  // namespace pq
