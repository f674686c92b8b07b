// bf16 instantiations of the tiled gather+MFMA kernel.
#include "pq_gemm_tiled_impl.cuh"
namespace pq {
hipError_t launch_gemm_tiled_bf16(const GemmArgs &a, hipStream_t stream) { return tiled_launch_at<bf16_t>(a, stream); }
}  // namespace pq
