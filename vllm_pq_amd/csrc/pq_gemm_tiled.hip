// Host side of the tiled gather+MFMA kernel (prefill): format checks and dtype dispatch.
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

hipError_t launch_gemm_tiled_bf16(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gemm_tiled_fp16(const GemmArgs &a, hipStream_t stream);

// Needs the PACKED code order, 16-byte aligned operands and a format the LDS image can hold.
bool gemm_tiled_supported(const GemmArgs &a) {
    if (!a.packed) return false;
    if (!packable(a.parts, a.G, a.d, a.K, a.M, a.code_bytes)) return false;
    // activations are staged in whole 64-feature tiles, codes consumed in register pieces of one or two
    // tiles (the tile shapes with several tiles per ring stage ask for more and are skipped when Kin does
    // not divide)
    {
        const int ks = ((16 / a.code_bytes) / a.M) * a.d / 8;          // k-steps per code chunk
        const int spp = ks >= 8 ? ks / 4 : ks >= 4 ? ks / 2 : ks;      // k-steps per piece
        if ((a.G * a.d) % (32 * spp)) return false;
    }
    if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_stride % 8)) return false;
    if ((reinterpret_cast<uintptr_t>(a.codes) & 15) || (reinterpret_cast<uintptr_t>(a.codebooks) & 15)) return false;
    // the epilogue stores 16 bytes (8 rows of one token) per lane
    if ((reinterpret_cast<uintptr_t>(a.y) & 15) || (a.y_stride % 8)) return false;
    // and reads the scales and bias of four consecutive rows with one 8-byte load
    if ((reinterpret_cast<uintptr_t>(a.scales) & 7) || (reinterpret_cast<uintptr_t>(a.bias) & 7)) return false;
    return true;
}

hipError_t launch_gemm_tiled(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? launch_gemm_tiled_bf16(a, stream) : launch_gemm_tiled_fp16(a, stream);
}

}  // namespace pq
