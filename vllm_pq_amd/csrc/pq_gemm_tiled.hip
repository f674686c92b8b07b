// Fused codebook gather + MFMA GEMM for large token counts (prefill) on gfx950.
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

bool gemm_tiled_supported(const GemmArgs &) { return false; }
hipError_t launch_gemm_tiled(const GemmArgs &, hipStream_t) { return hipErrorNotSupported; }

}  // namespace pq
