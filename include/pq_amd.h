/*
 * pq_amd.h -- C ABI of libpq_amd.so: the MI355X (gfx950) product-quantised linear layer.
 *
 * This is the boundary a host runtime binds (ctypes / cgo / JNI / a torch custom op).  It replaces,
 * for the PQ / AQLM tensor format, the native ops the reference registers in its torch extension:
 *
 *   pq_dequant  <->  aqlm_dequant(Tensor codes, Tensor codebooks, int[] codebook_partition_sizes)
 *                    csrc/torch_bindings.cpp:168-173, csrc/ops.h:128-131,
 *                    csrc/quantization/aqlm/gemm_kernels.cu:544-597, vllm/_custom_ops.py:360-371;
 *                    with scales != NULL it is dequantize_weight(codes, codebooks, scales) per
 *                    partition, vllm/model_executor/layers/quantization/aqlm.py:37-74, :105-126
 *   pq_gemm     <->  aqlm_gemm(Tensor input, Tensor codes, Tensor codebooks, Tensor scales,
 *                              int[] codebook_partition_sizes, Tensor? bias)
 *                    csrc/torch_bindings.cpp:160-166, csrc/ops.h:122-126,
 *                    csrc/quantization/aqlm/gemm_kernels.cu:519-542, vllm/_custom_ops.py:347-358;
 *                    numerically it is generic_dequantize_gemm, aqlm.py:94-127
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to memory the caller owns; the library never allocates,
 *     frees or keeps a pointer, and starts no threads.  `stream` is the caller's hipStream_t
 *     (NULL = the default stream); calls only enqueue work on it and never synchronise, so they
 *     can be captured into a HIP graph.
 *   - return value: PQ_OK (0) or a pq_status code; pq_last_error() gives a thread-local message.
 *     No C++ exception crosses the boundary.
 *   - fp16 / bf16 ("act dtype") selected by `dtype`; codebooks, scales, bias, x and y share it.
 *
 * Tensor layouts (row-major, contiguous unless a stride argument says otherwise), exactly the
 * parameters AQLMLinearMethod.create_weights registers (aqlm.py:253-316):
 *   codes      [N][G][M]      int8 (code_bytes 1) or int16 (code_bytes 2), signed storage;
 *                             index = value mod K (aqlm.py:32-34), K = 2^nbits
 *   codebooks  [P*M][K][1][d] act dtype; rows p*M .. p*M+M-1 belong to output partition p
 *   scales     [N][1][1][1]   act dtype, one per output row
 *   part_sizes [P]            HOST array: rows per output partition (q/k/v, gate/up), sum = N
 *   x          [T][Kin]       act dtype, Kin = G*d;  y [T][N] act dtype
 */
#ifndef PQ_AMD_H
#define PQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQ_ABI_VERSION 1
#define PQ_MAX_PARTS 8 /* the reference's kernels take at most 4 (gemm_kernels.cu:499-514) */

typedef enum pq_dtype { PQ_DTYPE_FP16 = 0, PQ_DTYPE_BF16 = 1 } pq_dtype;

typedef enum pq_status {
    PQ_OK = 0,
    PQ_ERR_INVALID_ARGUMENT = 1, /* shape / pointer / partition list rejected */
    PQ_ERR_UNSUPPORTED = 2,      /* a format this build has no kernel for */
    PQ_ERR_WORKSPACE = 3,        /* workspace too small */
    PQ_ERR_LAUNCH = 4            /* the HIP runtime reported an error at launch */
} pq_status;

/* capability bits of pq_caps() */
#define PQ_CAP_DEQUANT (1ull << 0)
#define PQ_CAP_GEMM (1ull << 1)
#define PQ_CAP_MFMA_SMALL_T (1ull << 2) /* fused gather+MFMA streaming kernel (decode) */
#define PQ_CAP_MFMA_TILED (1ull << 3)   /* fused gather+MFMA tiled kernel (prefill)   */
#define PQ_CAP_BF16 (1ull << 4)
#define PQ_CAP_FP16 (1ull << 5)
#define PQ_CAP_ELEMENTWISE (1ull << 6) /* pq_silu_and_mul, pq_rms_norm, pq_fused_add_rms_norm */
#define PQ_CAP_FUSED_SILU_MUL (1ull << 7) /* pq_gemm_packed_silu_mul runs fused for small batches */
#define PQ_CAP_EMBEDDING (1ull << 8)      /* pq_embedding */
#define PQ_CAP_FUSED_ADD_NORM (1ull << 9) /* pq_gemm_packed_add_norm / pq_gemm_packed_normed */

/* Version of this ABI (PQ_ABI_VERSION) and capability bits.  No reference counterpart. */
int pq_version(void);
uint64_t pq_caps(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *pq_last_error(void);

/*
 * Diagnostics only.  The dispatch knobs (environment variables PQ_AMD_FORCE_KERNEL, PQ_AMD_STREAM_CFG,
 * PQ_AMD_STREAM_KSPLIT, PQ_AMD_TILED_CFG, PQ_AMD_TILED_SLOTS, PQ_AMD_DECODE_CFG) are read once, when the library is
 * loaded, never per launch; tests and tuning tools that change them call this to have them read again.  Not to be
 * called while another thread is inside the library.  No reference counterpart.
 */
void pq_debug_reload_knobs(void);

/*
 * Reconstruct the dense weight  W[n][g*d + j] = sum_m codebooks[p(n)*M + m][idx(n,g,m)][j]
 * (x scales[n] when scales != NULL), out [N][G*d] in the act dtype.
 * M == 1 and scales == NULL is a pure gather and is bit-exact.  M > 1 sums in fp32 and rounds
 * once (nearest even); the scale multiply is one more rounding, as Tensor.mul does.
 * Replaces aqlm_dequant (gemm_kernels.cu:544-597) / dequantize_weight (aqlm.py:37-74).
 */
int pq_dequant(const void *codes, const void *codebooks, const void *scales /* nullable */,
               const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
               int32_t d, int32_t code_bytes, int32_t dtype, void *out, void *stream);

/*
 * Rows of that weight picked by an index vector: out[t][:] = W[ids[t]][:] for t < T, ids int64 on the device,
 * out rows out_row_stride elements apart.  Every element goes through pq_dequant's operations in its order, so
 * the result equals indexing pq_dequant's output bit for bit.  An index outside [0, N) gives a row of zeros.
 * It is QuantizeMethodBase.embedding (vllm/model_executor/layers/quantization/base_config.py:30-35), which
 * VocabParallelEmbedding.forward calls with indices already masked into the rank's shard
 * (vllm/model_executor/layers/vocab_parallel_embedding.py:403-422); the reference's unquantised method is
 * F.embedding (same file :45-47), and its AQLM method has no embedding().
 */
int pq_embedding(const int64_t *ids, int32_t T, const void *codes, const void *codebooks,
                 const void *scales /* nullable */, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                 int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *out, int64_t out_row_stride,
                 void *stream);

/*
 * Bytes of scratch with which pq_gemm / pq_gemm_packed may spread the K range of an output tile over several
 * blocks for this problem (0: they never do).  The workspace is optional: without it (NULL, or too small)
 * every problem still runs, on one block per tile.  It pays on layers with few rows and a long K at small
 * batches (Llama-3-8B down_proj, 4 to 64 tokens: up to 1.7x).
 * Contract: any 16-byte aligned device memory of at least that many bytes whose first 16 KiB are ZERO
 * before the first call that uses it; a call leaves them zero again, so one buffer can serve every later
 * call on the same stream.  Calls that may run concurrently (different streams) need a workspace each.
 * Results are deterministic (partial sums are added in a fixed order) but may differ in the last bit from
 * the same call without a workspace.  No reference counterpart (the reference allocates inside the op).
 */
size_t pq_gemm_workspace_bytes(int32_t T, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d,
                               int32_t code_bytes, int32_t dtype);
/*
 * The largest value pq_gemm_workspace_bytes returns for any problem (16 KiB of counters + 8 MiB of partial sums):
 * a buffer of this size, allocated once per stream that issues calls, serves every layer and batch size, so a host
 * never has to grow (and thereby replace) a workspace that launches already captured into a HIP graph point to.
 * A problem whose preferred split would need more runs with the largest split that fits.  No reference counterpart.
 */
size_t pq_gemm_workspace_max_bytes(void);

/*
 * y[t][n] = act( scales[n] * sum_k x[t][k] * Wunscaled[n][k] + bias[n] ), accumulated in fp32 with
 * one rounding at the end.  x rows are x_row_stride elements apart, y rows y_row_stride.
 * bias may be NULL.  Replaces aqlm_gemm (gemm_kernels.cu:519-542) and, for every format the
 * reference routes to PyTorch, generic_dequantize_gemm (aqlm.py:94-127).
 */
int pq_gemm(const void *x, int32_t T, int64_t x_row_stride, const void *codes, const void *codebooks,
            const void *scales, const void *bias /* nullable */, const int32_t *part_sizes, int32_t num_parts,
            int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y,
            int64_t y_row_stride, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Load-time re-layout of the codes into the order the MFMA kernels stream them in (one contiguous
 * KiB per 16-row fragment and k-chunk, see vllm_pq_amd/csrc/pq_gemm_stream.hip).  It is the hook
 * QuantizeMethodBase.process_weights_after_loading exists for
 * (vllm/model_executor/layers/quantization/base_config.py:37-42, called from
 * vllm/model_executor/model_loader/loader.py:157-168); the reference's AQLM method keeps the checkpoint
 * layout, so these three entry points have no counterpart there.
 *
 * pq_packed_codes_bytes: size of the packed buffer, or 0 when this format cannot be packed (a
 *   partition size that is not a multiple of 16, or a (d, K, M) the MFMA kernels do not take); the
 *   caller then keeps using pq_gemm on the reference layout.
 * pq_repack_codes: codes [N][G][M] -> packed (caller-allocated, 16-byte aligned, that many bytes); the packed
 *   buffer holds every code as its index (value mod K), which pq_gemm_packed relies on.
 * pq_gemm_packed: pq_gemm with `packed_codes` in place of `codes`; the same sums in the same order whenever
 *   both run the same kernel shape (the packed layout additionally unlocks the tiled kernel and the K split).
 */
size_t pq_packed_codes_bytes(const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M,
                             int32_t K, int32_t d, int32_t code_bytes);
int pq_repack_codes(const void *codes, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                    int32_t M, int32_t K, int32_t d, int32_t code_bytes, void *packed_codes, void *stream);
int pq_gemm_packed(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                   const void *codebooks, const void *scales, const void *bias /* nullable */,
                   const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                   int32_t d, int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride, void *workspace,
                   size_t workspace_bytes, void *stream);

/*
 * pq_gemm_packed for callers that cannot allocate device memory or bring a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): the K-split scratch is a buffer inside the
 * library's own image, 8 MiB + 16 KiB on every GPU the library runs on, held until the process ends.  The call
 * allocates and frees nothing.  Same arguments and results as pq_gemm_packed given a zeroed workspace of
 * pq_gemm_workspace_max_bytes(), except that a problem whose preferred K split needs more scratch than the buffer
 * holds (long K, more than 4096 rows and about 64 tokens or more) is rejected with PQ_ERR_WORKSPACE instead of run
 * on a smaller split.  All calls on one GPU share that GPU's buffer: calls that may overlap there (different
 * streams, different threads) race -- serialise them or use pq_gemm_packed with a workspace per stream.  The first
 * call on a device looks the buffer's address up (no allocation; do it once outside graph capture).
 * No counterpart in the reference: it is for callers that cannot allocate.
 */
int pq_gemm_packed_noalloc(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                           const void *codebooks, const void *scales, const void *bias /* nullable */,
                           const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                           int32_t d, int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride, void *stream);

/*
 * Which kernel pq_gemm (packed = 0) or pq_gemm_packed (packed = 1) would run for this problem: writes a short name ("mfma_stream", "mfma_tiled",
 * "generic") into buf.  For tests, benchmarks and profiles.  No reference counterpart.
 */
int pq_gemm_kernel_name(int32_t T, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                        int32_t dtype, const int32_t *part_sizes, int32_t num_parts, int32_t packed, char *buf,
                        size_t buf_len);

/*
 * The element-wise steps either side of the PQ linears in a decoder layer (vllm/model_executor/models/llama.py:
 * 89-93 act_fn after gate_up_proj, :272-284 the two norms before qkv_proj and gate_up_proj).  Rows are
 * row-stride elements apart; 16-byte aligned pointers and strides with a row length that is a multiple of 8
 * take the vectorised kernels, anything else a slower element-wise one.
 *
 * pq_silu_and_mul: out[t][j] = silu(x[t][j]) * x[t][I + j], x [T][2 I], out [T][I].  Replaces
 *   silu_and_mul(Tensor! out, Tensor input) (csrc/torch_bindings.cpp:55-56, csrc/activation_kernels.cu:22-38,
 *   :82-86); numerically SiluAndMul.forward_native (vllm/model_executor/layers/activation.py:73-76): silu in fp32
 *   rounded to the act dtype, the product rounded once more.
 * pq_rms_norm: out[t][:] = round(x[t][:] * rsqrt(mean(x[t]^2) + eps)) * weight, statistics in fp32.  Replaces
 *   rms_norm(Tensor! result, Tensor input, Tensor weight, float epsilon) (csrc/torch_bindings.cpp:106-109,
 *   csrc/layernorm_kernels.cu:17-44); numerically RMSNorm.forward_native (layers/layernorm.py:38-74).
 * pq_fused_add_rms_norm: z = x + residual in fp32; residual <- round(z); x <- the norm of z, both in place.
 *   Replaces fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon)
 *   (csrc/torch_bindings.cpp:112-115, csrc/layernorm_kernels.cu:50-100).  The variance uses the unrounded sum, as
 *   forward_native does (layernorm.py:45-48, :65); the reference's CUDA kernel uses the rounded one.
 */
int pq_silu_and_mul(const void *x, int64_t x_row_stride, int32_t T, int32_t I, int32_t dtype, void *out,
                    int64_t out_row_stride, void *stream);
int pq_rms_norm(const void *x, int64_t x_row_stride, const void *weight, float eps, int32_t T, int32_t H,
                int32_t dtype, void *out, int64_t out_row_stride, void *stream);
int pq_fused_add_rms_norm(void *x, int64_t x_row_stride, void *residual, int64_t res_row_stride, const void *weight,
                          float eps, int32_t T, int32_t H, int32_t dtype, void *stream);

/*
 * gate_up_proj with its activation: y[t][n] = silu(g[t][n]) * u[t][n], n < I, where [g | u] = pq_gemm_packed of a
 * layer with the two equal output partitions {I, I} (gate_proj, up_proj; M = 1) -- what LlamaMLP.forward computes
 * with gate_up_proj followed by act_fn (vllm/model_executor/models/llama.py:89-93).  The result has the roundings
 * of pq_gemm_packed followed by pq_silu_and_mul and is bit-identical to them wherever both run their sums in the same
 * order (always when the two steps are run; fused: when pq_gemm_packed takes the same kernel for the layer, as it
 * does for the layers the decode kernel fuses).  For small batches of the d = 8 / one-byte-code format the
 * activation runs in the epilogue of the decode kernel or of the streaming kernel (one launch, [T][2 I] never
 * written); pq_gemm_silu_mul_is_fused() says whether it does for a problem.  Otherwise the library runs the two steps
 * through `tmp`, caller-owned [T][2 I] in the act dtype (it may be NULL when the call is fused).
 * `workspace` as for pq_gemm_packed.  No reference counterpart (the reference always runs two ops).
 */
int pq_gemm_silu_mul_is_fused(int32_t T, int32_t I, int32_t G, int32_t K, int32_t d, int32_t code_bytes,
                              int32_t dtype);
int pq_gemm_packed_silu_mul(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                            const void *codebooks, const void *scales, const void *bias /* nullable */, int32_t I,
                            int32_t G, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y,
                            int64_t y_row_stride, void *tmp /* nullable */, void *workspace, size_t workspace_bytes,
                            void *stream);

/*
 * A fused add + RMSNorm between two PQ linears, carried by the two GEMMs either side of it -- the pair a decoder
 * layer runs as  x = o_proj(...); x, residual = post_attention_layernorm(x, residual); mlp(x)  and, across layers,
 * down_proj -> input_layernorm -> qkv_proj (vllm/model_executor/models/llama.py:272-284 with RMSNorm.forward's fused
 * add, vllm/model_executor/layers/layernorm.py:50-80).  Decode batches only (what the decode kernel takes: d = 8,
 * one codebook of up to 256 entries, one-byte packed codes, at most 128 tokens); the _supported() queries say
 * whether a problem runs, and a caller whose pair is not supported on both sides runs the three separate steps.
 * Not for a row-parallel layer whose output is summed over ranks first (tensor parallelism > 1).
 *
 *   pq_gemm_packed_add_norm: r = pq_gemm_packed(x, ...) + residual in fp32; `residual` [T][N] is overwritten with r
 *     (rounded once), xw [T][N] = r * norm_weight (rounded once), ssq [N / 16][Tp] (fp32, Tp = T rounded up to 16)
 *     receives the sum of r^2 over each 16-row fragment, per token.
 *   pq_gemm_packed_normed: y = pq_gemm_packed(xw, ...) with the accumulated sum multiplied by
 *     rsqrt(sum over fragments of ssq / hidden + eps) of its token before scale and bias; hidden = in_features.
 *
 * Against the reference's three steps the roundings differ: the reference rounds the first GEMM's output, then the
 * sum, then x / rms, then (x / rms) * w, each to the act dtype; here r is rounded once for the residual, r * w once,
 * and 1 / rms multiplies the second GEMM's fp32 sum.  The results agree to the accuracy of the act dtype (tests state
 * the tolerance); the new residual is the more accurate of the two.  No reference counterpart as one operator.
 */
int pq_gemm_packed_add_norm_supported(int32_t T, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                                      int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype);
int pq_gemm_packed_normed_supported(int32_t T, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                                    int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype);
int pq_gemm_packed_add_norm(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                            const void *codebooks, const void *scales, const void *bias /* nullable */,
                            const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                            int32_t d, int32_t code_bytes, int32_t dtype, void *residual, int64_t residual_row_stride,
                            const void *norm_weight, void *xw, int64_t xw_row_stride, float *ssq, void *workspace,
                            size_t workspace_bytes, void *stream);
int pq_gemm_packed_normed(const void *xw, int32_t T, int64_t xw_row_stride, const void *packed_codes,
                          const void *codebooks, const void *scales, const void *bias /* nullable */,
                          const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                          int32_t d, int32_t code_bytes, int32_t dtype, const float *ssq, float eps, void *y,
                          int64_t y_row_stride, void *workspace, size_t workspace_bytes, void *stream);

/*
 * pq_gemm_packed_normed for a gate_up layer with its activation: y [T][I] = silu(g) * u with [g | u] =
 * pq_gemm_packed_normed(xw, ...) of the layer's two equal partitions {I, I}; bit-identical to pq_gemm_packed_normed
 * followed by pq_silu_and_mul.  Runs where the decode kernel fuses the activation (K = 256 entries, in_features up to
 * 4096, at most 128 tokens): the _supported() query says so.  No reference counterpart as one operator.
 */
int pq_gemm_packed_normed_silu_mul_supported(int32_t T, int32_t I, int32_t G, int32_t K, int32_t d, int32_t code_bytes,
                                             int32_t dtype);
int pq_gemm_packed_normed_silu_mul(const void *xw, int32_t T, int64_t xw_row_stride, const void *packed_codes,
                                   const void *codebooks, const void *scales, const void *bias /* nullable */, int32_t I,
                                   int32_t G, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, const float *ssq,
                                   float eps, void *y, int64_t y_row_stride, void *stream);

/*
 * Small-message all-reduce (SUM) between the tensor-parallel ranks of one node, one process per GPU, over device
 * buffers the ranks map into each other (IPC handles; xGMI carries the loads and stores): the sum a row-parallel
 * PQ linear needs behind its GEMM (vllm/model_executor/layers/linear.py:1206-1213) at decode batch sizes.  It
 * replaces the reference's custom all-reduce for this path (csrc/custom_all_reduce.cuh:214-280 the one- and
 * two-stage kernels, :478-491 their dispatch by message size; host side
 * vllm/distributed/device_communicators/custom_all_reduce.py), with RCCL remaining the path for larger messages.
 *
 *   pq_ar_alloc      this rank's communication buffer (pq_ar_buffer_bytes() of uncached device memory, zeroed) and
 *                    its 64-byte IPC handle, which the host sends to the other ranks (any side channel)
 *   pq_ar_open       maps another rank's buffer from its handle; pq_ar_close / pq_ar_free undo the two
 *   pq_ar_create     a communicator from the `world` buffer addresses as seen by THIS process
 *                    (buffers[rank] = the own one), 2 to 8 ranks; pq_ar_destroy frees it (not the buffers)
 *   pq_ar_all_reduce out[i] = sum over ranks of in[i], i < count, dtype PQ_DTYPE_FP16 / PQ_DTYPE_BF16 or 2 = fp32;
 *                    fp32 accumulation in rank order, so every rank gets the same bits.  in == out is allowed.
 *                    The message must be a multiple of 16 bytes, 16-byte aligned, at most
 *                    pq_ar_max_message_bytes() (2 MiB); one shot up to 256 KiB, reduce-scatter + all-gather above.
 *                    Every rank must issue the same sequence of calls (count, dtype).  Enqueues one kernel on
 *                    `stream`; no allocation, no host synchronisation: capturable into a HIP graph.
 *   pq_ar_all_reduce_add_norm   the same sum (one shot) of a [T][H] fp16 / bf16 message with the fused residual add +
 *                    RMSNorm that follows a row-parallel layer in a decoder block
 *                    (vllm/model_executor/models/llama.py:272-284; layers/layernorm.py:38-74) in its epilogue:
 *                    z = round(sum over ranks of in) + residual; residual <- round(z); xw <- round(round(z / rms(z)) * w).
 *                    Bit-identical to pq_ar_all_reduce followed by pq_fused_add_rms_norm.  `in` is not modified and
 *                    may not alias residual or xw.  Rows of H % 8 == 0 up to 16384 elements, T * H * 2 <= 512 KiB:
 *                    pq_ar_add_norm_supported(T, H, dtype) says whether a problem is taken (larger batches run the
 *                    two steps).  Same calling rules as pq_ar_all_reduce.
 *   pq_ar_set_timeout_ms   how long a wait for a peer's flag lasts before it gives up (default 30 s)
 *   pq_ar_status     synchronous check after the stream has been synchronised: non-zero when a wait for a peer's
 *                    flag gave up.  A call whose wait gave up has filled its part of every output with NaN (all bits
 *                    set) instead of a sum, so the failure cannot go unnoticed downstream; the status stays set and
 *                    the communicator must not be used any more (the ranks' call counters may have come apart).
 * These calls are the only ones in the library that allocate (pq_ar_alloc) or keep pointers (the communicator).
 * Return 0 or a pq_status code; pq_ar_last_error() has the text.
 */
#define PQ_AR_MAX_RANKS 8
#define PQ_AR_IPC_HANDLE_BYTES 64
const char *pq_ar_last_error(void);
size_t pq_ar_buffer_bytes(void);
size_t pq_ar_max_message_bytes(void);
int pq_ar_alloc(void **buffer, void *ipc_handle /* out: PQ_AR_IPC_HANDLE_BYTES */);
int pq_ar_free(void *buffer);
int pq_ar_open(const void *ipc_handle, void **peer_buffer);
int pq_ar_close(void *peer_buffer);
int pq_ar_create(int32_t rank, int32_t world, void *const *buffers, void **comm);
int pq_ar_destroy(void *comm);
int pq_ar_all_reduce(void *comm, const void *in, void *out, int64_t count, int32_t dtype, void *stream);
int pq_ar_add_norm_supported(int32_t T, int32_t H, int32_t dtype);
int pq_ar_all_reduce_add_norm(void *comm, const void *in, int32_t T, int32_t H, int32_t dtype, void *residual,
                              const void *norm_weight, float eps, void *xw, void *stream);
int pq_ar_set_timeout_ms(void *comm, int64_t ms);
int pq_ar_status(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* PQ_AMD_H */
