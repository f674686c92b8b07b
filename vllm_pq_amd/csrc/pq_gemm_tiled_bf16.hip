// bf16 instantiations of the tiled gather+MFMA kernel.
#include "pq_gemm_tiled_impl.cuh"
namespace pq {
hipError_t launch_gemm_tiled_bf16(const GemmArgs &a, hipStream_t stream) { return tiled_launch_at<bf16_t>(a, stream); }
// what the launcher would do for this problem, for the name query: tile shape index and, when the launch is cut in two, the
// first row of the second launch (0: one launch)
void tiled_plan(const GemmArgs &a, int *cfg, int *split_row) {
    *split_row = choose_tiled_split(a);
    *cfg = *split_row > 0 ? 6 : choose_tiled_cfg(a);
}

}  // namespace pq
