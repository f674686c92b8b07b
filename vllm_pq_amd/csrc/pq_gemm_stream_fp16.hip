// fp16 instantiations of the streaming gather+MFMA kernel.
#include "pq_gemm_stream_impl.cuh"
namespace pq {
hipError_t launch_gemm_stream_fp16(const GemmArgs &a, hipStream_t stream) { return launch_at<fp16_t>(a, stream); }
hipError_t launch_gemm_stream_act_fp16(const GemmArgs &a, hipStream_t stream) { return launch_act_at<fp16_t>(a, stream); }
}  // namespace pq
