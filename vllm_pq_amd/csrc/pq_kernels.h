// Internal launch interface between the C ABI (pq_api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pq_common.cuh"

namespace pq {

struct DequantArgs {
    const void *codes;
    const void *codebooks;
    const void *scales;  // nullable
    Parts parts;
    int N, G, M, K, d, code_bytes, dtype;
    void *out;
};

// A fused add + RMSNorm between two linears, split over the two GEMMs either side of it (decode kernel only).
//   role 1, the GEMM in front: r = y + residual replaces the residual, xw = r * norm_w, ssq[fragment][t_pad] = the sum
//           of r^2 over the fragment's 16 rows, per token (t_pad = T rounded up to 16);
//   role 2, the GEMM behind: x is xw, and y is multiplied by rsqrt(sum over fragments of ssq / hidden + eps).
struct NormFuse {
    int role;
    void *residual;
    int64_t res_stride;
    const void *norm_w;
    void *xw;
    int64_t xw_stride;
    float *ssq;
    int hidden;
    float eps;
};

struct GemmArgs {
    const void *x;
    int T;
    int64_t x_stride;
    const void *codes;
    const void *codebooks;
    const void *scales;
    const void *bias;  // nullable
    Parts parts;
    int N, G, M, K, d, code_bytes, dtype;
    void *y;
    int64_t y_stride;
    void *workspace;
    size_t workspace_bytes;
    bool packed;  // codes are in the kernel-native PACKED order (pq_gemm_stream.hip)
    const NormFuse *norm = nullptr;   // the launch is one half of a fused add + RMSNorm (decode kernel only)
};

hipError_t launch_dequant(const DequantArgs &a, hipStream_t stream);
hipError_t launch_embed_rows(const DequantArgs &a, const int64_t *ids, int T, int64_t out_stride, hipStream_t stream);

// Any format: one wave per output row, VALU dot products.  Always available.
hipError_t launch_gemm_generic(const GemmArgs &a, hipStream_t stream);

// Fused gather + MFMA, weight-streaming (small T).  supported() says whether the format fits.
bool gemm_stream_supported(const GemmArgs &a);
// Tile shape of the streaming kernel for this problem: NF 16-row weight fragments x TF 16-token fragments
// per block.  *model_us (optional) receives the modelled launch duration for T > 16, 0 otherwise.
struct StreamCfg { int nf, tf, ksplit; };
// Workspace layout of the K-split launches: WS_COUNTER_BYTES of per-tile arrival counters (zero between
// launches), then the fp32 partial sums [ksplit][T][N].
constexpr size_t WS_COUNTER_BYTES = 16384;
constexpr size_t WS_PARTIAL_BYTES = (size_t)8 << 20;
// Whether `ksplit` blocks may share one tile's K range in the streaming kernel for this problem and tile shape
// (needs packed codes, a workspace that holds the partial sums, at most 4096 tiles).
bool stream_ksplit_ok(const GemmArgs &a, int nf, int tf, int ksplit);
// Bytes of workspace with which this problem may run K-split (0: it never does).
size_t stream_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes);
// The same without the cap of WS_PARTIAL_BYTES: what the problem's preferred split would take.
size_t stream_preferred_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes);
StreamCfg choose_stream_cfg(const GemmArgs &a, double *model_us);
hipError_t launch_gemm_stream(const GemmArgs &a, hipStream_t stream);
// Fused gate/up GEMM + SiLU-and-mul on the streaming kernel: a.parts = two equal partitions (gate, up), a.y is
// [T][N / 2].  supported(): d = 8, one codebook, one-byte codes, packed, partitions a multiple of 16 rows, and a
// batch small enough that the fused launch beats the tiled GEMM followed by the stand-alone activation.
bool gemm_stream_act_supported(const GemmArgs &a);
int choose_stream_act_nf(const GemmArgs &a);
hipError_t launch_gemm_stream_act(const GemmArgs &a, hipStream_t stream);

// Decode batches (16 tokens per block, up to 128 as groups over the grid; d = 8, one codebook of up to 256 entries, one-byte packed codes, whole
// 512-feature chunks): pq_gemm_decode_impl.cuh.  `waves` per block, of which 1 << wk_log2 along K; nf fragments
// and c chunks per wave.  nf == 0: not taken.
struct DecodeGeom { int nf, wk_log2, c, waves, ksplit, cpb; };   // ksplit blocks per tile along K, cpb chunks each
// Bytes of workspace with which a decode batch of this problem runs K-split over blocks (0: K fits one block, or
// the decode kernel does not take the format).
size_t decode_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes);
bool gemm_decode_supported(const GemmArgs &a);
bool gemm_decode_preferred(const GemmArgs &a);
double decode_model_us(const GemmArgs &a);   // modelled launch duration (0: not taken), for batches beyond 16 tokens
DecodeGeom choose_decode_geom(const GemmArgs &a);
hipError_t launch_gemm_decode(const GemmArgs &a, hipStream_t stream);
// Whether launch_gemm_decode takes this problem with a.norm set (role 1 or 2).
bool gemm_decode_norm_supported(const GemmArgs &a);
// gate_up_proj with SiLU-and-mul in the decode kernel's epilogue: a.parts = two equal partitions (gate, up), a.y is
// [T][N / 2]; a.norm may name role 2 (the layer sits behind a fused add + RMSNorm).  supported(): what the decode
// kernel takes, K = 256 entries, at most 8 chunks (no K split), blocks of 8 waves.
bool gemm_decode_act_supported(const GemmArgs &a);
hipError_t launch_gemm_decode_act(const GemmArgs &a, hipStream_t stream);

// Mid-size batches (32 tokens per block over gridDim.z, any number of rows per block in rounds; d = 8, one codebook of
// up to 256 entries, one-byte packed codes, at most 8 chunks of 512 features): pq_gemm_wide_impl.cuh.
struct WideGeom { int nf, rounds, wk_log2; };   // fragments per wave and round, rounds per block, log2 of the waves along K
bool gemm_wide_supported(const GemmArgs &a);
WideGeom choose_wide_geom(const GemmArgs &a, double *model_us);   // *model_us (optional): modelled launch duration
hipError_t launch_gemm_wide(const GemmArgs &a, hipStream_t stream);

// Load-time re-layout of the codes for the MFMA kernels.
bool stream_format_supported(int d, int K, int M, int code_bytes);
bool packable(const Parts &parts, int G, int d, int K, int M, int code_bytes);
size_t packed_bytes(int N, int G, int M, int code_bytes);
hipError_t launch_repack(const void *codes, int N, int G, int M, int K, int d, int code_bytes, void *packed,
                         hipStream_t stream);

// Fused gather + MFMA, LDS-tiled activations (large T).
bool gemm_tiled_supported(const GemmArgs &a);
hipError_t launch_gemm_tiled(const GemmArgs &a, hipStream_t stream);
void tiled_plan(const GemmArgs &a, int *cfg, int *split_row);   // tile shape index; first row of a second launch (0: none)

// The element-wise neighbours of the linears (pq_elementwise.hip).  H = row length (for silu_and_mul: the
// output's, the input row holds 2 H); residual != NULL selects the fused add.
struct ElementwiseArgs {
    const void *x;
    int64_t x_stride;
    void *residual;
    int64_t res_stride;
    const void *weight;
    float eps;
    int T, H, dtype;
    void *out;
    int64_t out_stride;
};
constexpr int RMS_SCALAR_MAX_H = 15360;   // rows the unaligned fallback holds in LDS as fp32
bool rms_norm_vectorised(const ElementwiseArgs &a);
hipError_t launch_silu_and_mul(const ElementwiseArgs &a, hipStream_t stream);
hipError_t launch_rms_norm(const ElementwiseArgs &a, hipStream_t stream);

}  // namespace pq
