// Geometry shared by the gather+MFMA kernels and the code re-layout: how a lane's 16-byte code load
// maps to MFMA k-steps, and how large the replicated LDS codebook image is.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace pq {

constexpr int ST_MAX_LDS_CB = 64 * 1024;   // LDS budget of the replicated codebook image
// LDS offset of the streaming kernel's activation slots: behind the codebook image and the reduction area (which
// share the front of the LDS), 16-byte aligned
__host__ __device__ inline uint32_t stream_x_lds_offset(size_t image_bytes, size_t reduce_bytes) {
    const size_t m = image_bytes > reduce_bytes ? image_bytes : reduce_bytes;
    return (uint32_t)((m + 15) & ~(size_t)15);
}

template <int D>
struct StreamGeom {
    static constexpr int EB = 2 * D;               // bytes per codebook entry
    static constexpr int RMAX = D == 8 ? 16 : 32;  // copies for a conflict-free gather
    static constexpr int RMIN = 16 / EB;           // the image is written in 16-byte slots
};

// Replication factor: largest power of two <= RMAX with M*K*EB*R <= 64 KiB (0: does not fit).
template <int D>
__host__ __device__ inline int stream_replication(int K, int M) {
    int r = StreamGeom<D>::RMAX;
    while (r >= StreamGeom<D>::RMIN && (size_t)M * K * StreamGeom<D>::EB * r > (size_t)ST_MAX_LDS_CB) r >>= 1;
    return r >= StreamGeom<D>::RMIN ? r : 0;
}

// CB = bytes per stored code, D = sub-vector length, M = codebooks.
template <int CB, int D, int M>
struct StreamFmt {
    static constexpr int CPL = 16 / CB;      // codes per lane per 16-byte load
    static constexpr int GPL = CPL / M;      // input groups per lane per load
    static constexpr int CG = 4 * GPL;       // input groups per chunk (4 lane quads)
    static constexpr int KS = GPL * D / 8;   // MFMA k-steps (32 k each) per chunk
    static constexpr int CPS = 8 / D;        // groups per k-step per lane
    static_assert(CPL % M == 0 && KS >= 1, "format does not fill an MFMA k-step");
};

// Runtime twin of StreamFmt::CG for host code.
inline int chunk_groups(int code_bytes, int M) { return 4 * ((16 / code_bytes) / M); }

}  // namespace pq
