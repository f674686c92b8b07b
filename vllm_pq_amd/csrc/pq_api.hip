// C ABI of libpq_amd.so (declared in include/pq_amd.h): argument checking, kernel selection,
// error reporting.  No allocation, no synchronisation, no global mutable state besides the
// thread-local error text.
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pq_kernels.h"
#include "pq_knobs.h"

namespace {

thread_local char g_err[512] = "";

int fail(pq_status code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return (int)code;
}

bool is_pow2(int v) { return v >= 2 && (v & (v - 1)) == 0; }

// Common format checks; mirrors the reference's own preconditions:
//   K a power of two            (nbits = K.bit_length() - 1, aqlm.py:86)
//   sum(part_sizes) == N        (aqlm.py:111-112)
//   codes wide enough for K     (get_int_dtype, aqlm.py:20-29)
int check_format(const char *fn, const int32_t *part_sizes, int num_parts, int N, int G, int M, int K, int d,
                 int code_bytes, int dtype, pq::Parts *out) {
    if (num_parts < 1 || num_parts > PQ_MAX_PARTS)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: num_parts=%d outside [1, %d]", fn, num_parts, PQ_MAX_PARTS);
    if (!part_sizes) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: part_sizes is NULL", fn);
    if (N < 0 || G < 0 || M < 1 || d < 1)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: bad shape N=%d G=%d M=%d d=%d", fn, N, G, M, d);
    if (!is_pow2(K)) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: K=%d is not a power of two >= 2", fn, K);
    if (code_bytes != 1 && code_bytes != 2)
        return fail(PQ_ERR_UNSUPPORTED, "%s: code_bytes=%d (int8 and int16 codes are supported)", fn, code_bytes);
    if ((code_bytes == 1 && K > 256) || (code_bytes == 2 && K > 65536))
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: K=%d does not fit %d-byte codes", fn, K, code_bytes);
    if (dtype != PQ_DTYPE_FP16 && dtype != PQ_DTYPE_BF16)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: dtype=%d is neither fp16 (0) nor bf16 (1)", fn, dtype);
    if ((int64_t)G * d > INT32_MAX) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: in_features overflow", fn);
    long long tot = 0;
    out->num = num_parts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) out->size[p] = 0;
    for (int p = 0; p < num_parts; ++p) {
        if (part_sizes[p] < 0) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: part_sizes[%d]=%d", fn, p, part_sizes[p]);
        out->size[p] = part_sizes[p];
        tot += part_sizes[p];
    }
    if (tot != N)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: sum(part_sizes)=%lld != N=%d", fn, tot, N);
    return PQ_OK;
}

enum Kernel { K_GENERIC = 0, K_STREAM = 1, K_TILED = 2, K_DECODE = 3, K_WIDE = 4 };

Kernel choose(const pq::GemmArgs &a) {
    const int force = pq::knobs().force_kernel;
    if (force == K_GENERIC) return K_GENERIC;
    if (force == K_STREAM && pq::gemm_stream_supported(a)) return K_STREAM;
    if (force == K_TILED && pq::gemm_tiled_supported(a)) return K_TILED;
    if (force == K_DECODE && pq::gemm_decode_supported(a)) return K_DECODE;
    if (force == K_WIDE && pq::gemm_wide_supported(a)) return K_WIDE;
    if (force < 0 && pq::gemm_decode_preferred(a)) return K_DECODE;
    // Mid-size batches of 5 to 20 GFLOP on layers of at least five 512-feature chunks (every wave of a block gets a
    // chunk): the mid-batch kernel.  Measured on the MI355X against every tile shape of the other kernels
    // (tools/autotune.py --full): ahead by 7 - 18 % in that range -- blocks of 8 to 16 units a wave, one pass over the chip
    // --, behind the decode kernel's token groups below it (a launch is then mostly its fixed phases, and 16-token blocks
    // take in half the activations) and behind the tiled kernel above it (long main loops at a higher MFMA rate).
    if (force < 0 && a.T > 16 && a.G / 64 >= 5 && pq::gemm_wide_supported(a)) {
        const double flops = 2.0 * (double)a.T * (double)a.N * (double)a.G * (double)a.d;
        if (flops >= 4.5e9 && flops <= 2.1e10) return K_WIDE;
    }
    const bool st = pq::gemm_stream_supported(a);
    const bool ti = pq::gemm_tiled_supported(a);
    if (ti && st && a.T > 16) {
        // Mid-size batches.  The tiled kernel does not split K, so below 256 output tiles of 64 x 64 it
        // cannot fill the 256 CUs and one launch lasts about 3 us + 3 us per 1024 input features (one
        // 64 x 64 tile per CU, measured: the HBM latency of the code loads paces the short steps); the
        // streaming kernel splits K over its waves, the decode kernel runs the batch as groups of 16 tokens:
        // whichever has the shortest modelled duration.
        const double out_tiles = (double)a.T * a.N / 4096.0;
        if (out_tiles <= 256.0) {
            double stream_us = 0.0;
            pq::choose_stream_cfg(a, &stream_us);
            const long tiles64 = (long)((a.N + 63) / 64) * ((a.T + 63) / 64);
            // sub-vectors shorter than 8 take 8/d gathers per operand and two-byte codes a less replicated
            // codebook image: measured 1.15x ... 2.6x the time of d = 8 with one-byte codes
            // (d = 2 with two-byte codes: four gathers per operand from an image with one copy per entry, 4x)
            const double fmt = (a.d == 2 && a.code_bytes == 2) ? 4.0 : sqrt(8.0 / a.d) * (a.code_bytes == 2 ? 1.3 : 1.0);
            const double tiled_us = (double)((tiles64 + 255) / 256) * (3.0 + 3.0 * (double)a.G * a.d / 1024.0) * fmt;
            const double decode_us = (force < 0 && pq::gemm_decode_supported(a)) ? pq::decode_model_us(a) : 0.0;
            const double other_us = (stream_us > 0.0 && stream_us <= tiled_us) ? stream_us : tiled_us;
            if (decode_us > 0.0 && decode_us < other_us) return K_DECODE;
            if (stream_us > 0.0 && stream_us <= tiled_us) return K_STREAM;
        }
        return K_TILED;
    }
    if (ti && !st) return K_TILED;
    if (st) return K_STREAM;
    return K_GENERIC;
}

}  // namespace

namespace pq {

static Knobs read_knobs() {
    Knobs k{-1, 0, 0, 0, -1, 0, -1, 0, 0, 0, 0};
    const char *e = getenv("PQ_AMD_FORCE_KERNEL");
    if (e && *e) {
        if (!strcmp(e, "generic")) k.force_kernel = K_GENERIC;
        else if (!strcmp(e, "stream")) k.force_kernel = K_STREAM;
        else if (!strcmp(e, "tiled")) k.force_kernel = K_TILED;
        else if (!strcmp(e, "decode")) k.force_kernel = K_DECODE;
        else if (!strcmp(e, "wide")) k.force_kernel = K_WIDE;
    }
    e = getenv("PQ_AMD_STREAM_CFG");
    if (e && *e && sscanf(e, "%dx%d", &k.stream_nf, &k.stream_tf) != 2) k.stream_nf = k.stream_tf = 0;
    e = getenv("PQ_AMD_STREAM_KSPLIT");
    if (e && *e) k.stream_ksplit = atoi(e);
    e = getenv("PQ_AMD_TILED_CFG");
    if (e && *e) k.tiled_cfg = atoi(e);
    e = getenv("PQ_AMD_TILED_SLOTS");
    if (e && *e) k.tiled_slots = atoi(e);
    e = getenv("PQ_AMD_TILED_SPLIT");
    if (e && *e) k.tiled_split = atoi(e);
    e = getenv("PQ_AMD_DECODE_CFG");
    if (e && *e && sscanf(e, "%dx%d", &k.decode_waves, &k.decode_nf) != 2) k.decode_waves = k.decode_nf = 0;
    e = getenv("PQ_AMD_WIDE_CFG");
    if (e && *e && sscanf(e, "%dx%d", &k.wide_nf, &k.wide_rounds) != 2) k.wide_nf = k.wide_rounds = 0;
    return k;
}
static Knobs g_knobs = read_knobs();   // at library load
const Knobs &knobs() { return g_knobs; }
void reload_knobs() { g_knobs = read_knobs(); }

}  // namespace pq

extern "C" {

void pq_debug_reload_knobs(void) { pq::reload_knobs(); }

int pq_version(void) { return PQ_ABI_VERSION; }

uint64_t pq_caps(void) {
    return PQ_CAP_DEQUANT | PQ_CAP_GEMM | PQ_CAP_MFMA_SMALL_T | PQ_CAP_MFMA_TILED | PQ_CAP_BF16 | PQ_CAP_FP16 |
           PQ_CAP_ELEMENTWISE | PQ_CAP_FUSED_SILU_MUL | PQ_CAP_EMBEDDING | PQ_CAP_FUSED_ADD_NORM;
}

const char *pq_last_error(void) { return g_err; }

int pq_dequant(const void *codes, const void *codebooks, const void *scales, const int32_t *part_sizes,
               int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
               int32_t dtype, void *out, void *stream) {
    pq::DequantArgs a;
    int rc = check_format("pq_dequant", part_sizes, num_parts, N, G, M, K, d, code_bytes, dtype, &a.parts);
    if (rc) return rc;
    if (N == 0 || G == 0) return PQ_OK;
    if (!codes || !codebooks || !out) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_dequant: NULL tensor pointer");
    a.codes = codes; a.codebooks = codebooks; a.scales = scales;
    a.N = N; a.G = G; a.M = M; a.K = K; a.d = d; a.code_bytes = code_bytes; a.dtype = dtype; a.out = out;
    hipError_t e = pq::launch_dequant(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "pq_dequant: %s", hipGetErrorString(e));
    return PQ_OK;
}

int pq_embedding(const int64_t *ids, int32_t T, const void *codes, const void *codebooks, const void *scales,
                 const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d,
                 int32_t code_bytes, int32_t dtype, void *out, int64_t out_row_stride, void *stream) {
    pq::DequantArgs a;
    int rc = check_format("pq_embedding", part_sizes, num_parts, N, G, M, K, d, code_bytes, dtype, &a.parts);
    if (rc) return rc;
    if (T < 0) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_embedding: T=%d", T);
    if (T == 0 || G == 0) return PQ_OK;
    if (out_row_stride < (int64_t)G * d) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_embedding: out_row_stride %lld < %d features", (long long)out_row_stride, G * d);
    if (!ids || !codes || !codebooks || !out) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_embedding: NULL tensor pointer");
    a.codes = codes; a.codebooks = codebooks; a.scales = scales;
    a.N = N; a.G = G; a.M = M; a.K = K; a.d = d; a.code_bytes = code_bytes; a.dtype = dtype; a.out = out;
    hipError_t e = pq::launch_embed_rows(a, ids, T, out_row_stride, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "pq_embedding: %s", hipGetErrorString(e));
    return PQ_OK;
}

size_t pq_gemm_workspace_bytes(int32_t T, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                               int32_t /*dtype*/) {
    if (T <= 0 || N <= 0 || G <= 0) return 0;
    const size_t st = pq::stream_workspace_bytes(T, N, G, M, K, d, code_bytes);
    const size_t dc = pq::decode_workspace_bytes(T, N, G, M, K, d, code_bytes);
    return st > dc ? st : dc;
}

size_t pq_gemm_workspace_max_bytes(void) { return pq::WS_COUNTER_BYTES + pq::WS_PARTIAL_BYTES; }

static int fill_gemm(pq::GemmArgs &a, const char *fn, const void *x, int32_t T, int64_t x_row_stride,
                     const void *codes, const void *codebooks, const void *scales, const void *bias,
                     const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                     int32_t d, int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride) {
    int rc = check_format(fn, part_sizes, num_parts, N, G, M, K, d, code_bytes, dtype, &a.parts);
    if (rc) return rc;
    if (T < 0) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: T=%d", fn, T);
    if (x_row_stride < (int64_t)G * d)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: x_row_stride=%lld < in_features=%lld", fn, (long long)x_row_stride,
                    (long long)G * d);
    if (y_row_stride < N)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: y_row_stride=%lld < N=%d", fn, (long long)y_row_stride, N);
    a.x = x; a.T = T; a.x_stride = x_row_stride; a.codes = codes; a.codebooks = codebooks; a.scales = scales;
    a.bias = bias; a.N = N; a.G = G; a.M = M; a.K = K; a.d = d; a.code_bytes = code_bytes; a.dtype = dtype;
    a.y = y; a.y_stride = y_row_stride; a.workspace = nullptr; a.workspace_bytes = 0; a.packed = false;
    return PQ_OK;
}

static int run_gemm(const char *fn, bool packed, const void *x, int32_t T, int64_t x_row_stride, const void *codes,
                    const void *codebooks, const void *scales, const void *bias, const int32_t *part_sizes,
                    int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                    int32_t dtype, void *y, int64_t y_row_stride, void *workspace, size_t workspace_bytes,
                    void *stream, bool clear_counters = false) {
    pq::GemmArgs a;
    int rc = fill_gemm(a, fn, x, T, x_row_stride, codes, codebooks, scales, bias, part_sizes, num_parts, N, G, M, K,
                       d, code_bytes, dtype, y, y_row_stride);
    if (rc) return rc;
    if (T == 0 || N == 0) return PQ_OK;
    if (!y || !scales) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    if (G > 0 && (!x || !codes || !codebooks)) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
    a.packed = packed;
    if (packed && !pq::packable(a.parts, G, d, K, M, code_bytes))
        return fail(PQ_ERR_UNSUPPORTED, "%s: this format has no packed layout (pq_packed_codes_bytes() == 0)", fn);
    const Kernel k = choose(a);
    if (packed && k == K_GENERIC)
        return fail(PQ_ERR_INVALID_ARGUMENT,
                    "%s: x, codebooks and packed_codes must be 16-byte aligned and x_row_stride a multiple of 8", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (clear_counters && ((k == K_STREAM && pq::choose_stream_cfg(a, nullptr).ksplit > 1) ||
                           (k == K_DECODE && pq::choose_decode_geom(a).ksplit > 1))) {
        // a workspace other calls have used: the arrival counters at its head start this launch at zero
        e = hipMemsetAsync(workspace, 0, pq::WS_COUNTER_BYTES, s);
        if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: clearing the scratch counters: %s", fn, hipGetErrorString(e));
    }
    switch (k) {
    case K_TILED: e = pq::launch_gemm_tiled(a, s); break;
    case K_STREAM: e = pq::launch_gemm_stream(a, s); break;
    case K_DECODE: e = pq::launch_gemm_decode(a, s); break;
    case K_WIDE: e = pq::launch_gemm_wide(a, s); break;
    default: e = pq::launch_gemm_generic(a, s); break;
    }
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(e));
    return PQ_OK;
}

int pq_gemm(const void *x, int32_t T, int64_t x_row_stride, const void *codes, const void *codebooks,
            const void *scales, const void *bias, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
            int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride,
            void *workspace, size_t workspace_bytes, void *stream) {
    return run_gemm("pq_gemm", false, x, T, x_row_stride, codes, codebooks, scales, bias, part_sizes, num_parts, N, G,
                    M, K, d, code_bytes, dtype, y, y_row_stride, workspace, workspace_bytes, stream);
}

int pq_gemm_packed(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes, const void *codebooks,
                   const void *scales, const void *bias, const int32_t *part_sizes, int32_t num_parts, int32_t N,
                   int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y,
                   int64_t y_row_stride, void *workspace, size_t workspace_bytes, void *stream) {
    return run_gemm("pq_gemm_packed", true, x, T, x_row_stride, packed_codes, codebooks, scales, bias, part_sizes,
                    num_parts, N, G, M, K, d, code_bytes, dtype, y, y_row_stride, workspace, workspace_bytes, stream);
}

// ---- pq_gemm_packed for callers that cannot allocate ----------------------------------------------------------
// The K-split scratch lives in the library's own image, one copy per GPU.  `scratch` is only the host-side handle
// of the symbol; the device address is asked for once per device and that address is what launches get.
constexpr size_t SCRATCH_BYTES = pq::WS_COUNTER_BYTES + pq::WS_PARTIAL_BYTES;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
static std::atomic<void *> g_scratch_ptr[SCRATCH_MAX_DEVICES];

// Device address of `scratch` on the current device (NULL and an error text on failure).
static void *scratch_of_current_device(const char *fn) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { fail(PQ_ERR_LAUNCH, "%s: hipGetDevice: %s", fn, hipGetErrorString(e)); return nullptr; }
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) { fail(PQ_ERR_UNSUPPORTED, "%s: device %d (at most %d devices)", fn, dev, SCRATCH_MAX_DEVICES); return nullptr; }
    void *ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (ptr) return ptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        fail(PQ_ERR_LAUNCH, "%s: hipGetSymbolAddress(scratch) on device %d: %s", fn, dev, hipGetErrorString(e));
        return nullptr;
    }
    g_scratch_ptr[dev].store(ptr, std::memory_order_release);   // every thread resolves the same address
    return ptr;
}

int pq_gemm_packed_noalloc(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                           const void *codebooks, const void *scales, const void *bias, const int32_t *part_sizes,
                           int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                           int32_t dtype, void *y, int64_t y_row_stride, void *stream) {
    const char *fn = "pq_gemm_packed_noalloc";
    void *ptr = nullptr;
    if (T > 0 && N > 0 && G > 0) {
        size_t want = pq::stream_preferred_workspace_bytes(T, N, G, M, K, d, code_bytes);
        const size_t want_dc = pq::decode_workspace_bytes(T, N, G, M, K, d, code_bytes);
        if (want_dc > want) want = want_dc;
        if (want > SCRATCH_BYTES)
            return fail(PQ_ERR_WORKSPACE, "%s: T=%d x N=%d with %d input groups splits K with %zu bytes of scratch, the "
                        "built-in buffer holds %zu: use pq_gemm_packed with a workspace of the caller's, or a smaller batch",
                        fn, T, N, G, want, SCRATCH_BYTES);
        ptr = scratch_of_current_device(fn);
        if (!ptr) return PQ_ERR_LAUNCH;
    }
    return run_gemm(fn, true, x, T, x_row_stride, packed_codes, codebooks, scales, bias, part_sizes, num_parts, N, G, M,
                    K, d, code_bytes, dtype, y, y_row_stride, ptr, ptr ? SCRATCH_BYTES : 0, stream, true);
}

size_t pq_packed_codes_bytes(const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K,
                             int32_t d, int32_t code_bytes) {
    pq::Parts parts;
    if (check_format("pq_packed_codes_bytes", part_sizes, num_parts, N, G, M, K, d, code_bytes, PQ_DTYPE_BF16, &parts))
        return 0;
    if (!pq::packable(parts, G, d, K, M, code_bytes)) return 0;
    return pq::packed_bytes(N, G, M, code_bytes);
}

int pq_repack_codes(const void *codes, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M,
                    int32_t K, int32_t d, int32_t code_bytes, void *packed_codes, void *stream) {
    pq::Parts parts;
    int rc = check_format("pq_repack_codes", part_sizes, num_parts, N, G, M, K, d, code_bytes, PQ_DTYPE_BF16, &parts);
    if (rc) return rc;
    if (!pq::packable(parts, G, d, K, M, code_bytes))
        return fail(PQ_ERR_UNSUPPORTED, "pq_repack_codes: this format has no packed layout");
    if (N == 0 || G == 0) return PQ_OK;
    if (!codes || !packed_codes) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_repack_codes: NULL tensor pointer");
    if (reinterpret_cast<uintptr_t>(packed_codes) & 15)
        return fail(PQ_ERR_INVALID_ARGUMENT, "pq_repack_codes: packed_codes must be 16-byte aligned");
    hipError_t e = pq::launch_repack(codes, N, G, M, K, d, code_bytes, packed_codes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "pq_repack_codes: %s", hipGetErrorString(e));
    return PQ_OK;
}

int pq_gemm_kernel_name(int32_t T, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                        int32_t dtype, const int32_t *part_sizes, int32_t num_parts, int32_t packed, char *buf,
                        size_t buf_len) {
    pq::GemmArgs a;
    // aligned dummy pointers: the choice depends on alignment, which real tensors from an allocator have
    void *al = reinterpret_cast<void *>(uintptr_t(256));
    int rc = fill_gemm(a, "pq_gemm_kernel_name", al, T, (int64_t)G * d, al, al, al, nullptr, part_sizes, num_parts, N,
                       G, M, K, d, code_bytes, dtype, al, N);
    if (rc) return rc;
    if (!buf || buf_len == 0) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_gemm_kernel_name: no buffer");
    a.packed = packed != 0;
    // the choice for a launch that brings the full-size workspace, as a loaded layer's launches do
    a.workspace = al;
    a.workspace_bytes = pq::WS_COUNTER_BYTES + pq::WS_PARTIAL_BYTES;
    if (a.packed && !pq::packable(a.parts, G, d, K, M, code_bytes))
        return fail(PQ_ERR_UNSUPPORTED, "pq_gemm_kernel_name: this format has no packed layout");
    static const char *names[] = {"generic", "mfma_stream", "mfma_tiled", "mfma_decode", "mfma_wide"};
    const Kernel k = choose(a);
    int n = snprintf(buf, buf_len, "%s%s", names[k], a.packed ? "_packed" : "");
    // a caller with a long buffer (tuning tools) also gets the launch geometry behind the name
    if (buf_len >= 128 && n > 0 && (size_t)n < buf_len) {
        if (k == K_STREAM) {
            double us = 0.0;
            const pq::StreamCfg c = pq::choose_stream_cfg(a, &us);
            snprintf(buf + n, buf_len - n, " %dx%d ks%d model %.1f us", c.nf, c.tf, c.ksplit, us);
        } else if (k == K_TILED) {
            int cfg = 0, split = 0;
            pq::tiled_plan(a, &cfg, &split);
            if (split > 0) snprintf(buf + n, buf_len - n, " cfg%d rows < %d, cfg1 from there", cfg, split);
            else snprintf(buf + n, buf_len - n, " cfg%d", cfg);
        } else if (k == K_WIDE) {
            double us = 0.0;
            const pq::WideGeom g = pq::choose_wide_geom(a, &us);
            snprintf(buf + n, buf_len - n, " %dx%d model %.1f us", g.nf, g.rounds, us);
        } else if (k == K_DECODE) {
            const pq::DecodeGeom g = pq::choose_decode_geom(a);
            snprintf(buf + n, buf_len - n, " %dx%d ks%d model %.1f us", g.waves, g.nf, g.ksplit, pq::decode_model_us(a));
        }
    }
    return PQ_OK;
}

// ---- element-wise neighbours ---------------------------------------------------------------------
static int check_rows(const char *fn, int32_t T, int32_t H, int32_t dtype) {
    if (T < 0 || H < 0) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: bad shape T=%d H=%d", fn, T, H);
    if (dtype != PQ_DTYPE_FP16 && dtype != PQ_DTYPE_BF16)
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: dtype=%d is neither fp16 (0) nor bf16 (1)", fn, dtype);
    return PQ_OK;
}

int pq_silu_and_mul(const void *x, int64_t x_row_stride, int32_t T, int32_t I, int32_t dtype, void *out,
                    int64_t out_row_stride, void *stream) {
    int rc = check_rows("pq_silu_and_mul", T, I, dtype);
    if (rc) return rc;
    if (T == 0 || I == 0) return PQ_OK;
    if (!x || !out) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_silu_and_mul: NULL tensor pointer");
    if (x_row_stride < 2 * (int64_t)I || out_row_stride < I)
        return fail(PQ_ERR_INVALID_ARGUMENT, "pq_silu_and_mul: row strides %lld / %lld shorter than the rows (2*%d / %d)",
                    (long long)x_row_stride, (long long)out_row_stride, I, I);
    pq::ElementwiseArgs a{x, x_row_stride, nullptr, 0, nullptr, 0.f, T, I, dtype, out, out_row_stride};
    hipError_t e = pq::launch_silu_and_mul(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "pq_silu_and_mul: %s", hipGetErrorString(e));
    return PQ_OK;
}

static int rms_common(const char *fn, const void *x, int64_t x_row_stride, void *residual, int64_t res_row_stride,
                      const void *weight, float eps, int32_t T, int32_t H, int32_t dtype, void *out,
                      int64_t out_row_stride, void *stream) {
    int rc = check_rows(fn, T, H, dtype);
    if (rc) return rc;
    if (T == 0 || H == 0) return PQ_OK;
    if (!x || !out || !weight) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    if (x_row_stride < H || out_row_stride < H || (residual && res_row_stride < H))
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: a row stride is shorter than the row (%d)", fn, H);
    pq::ElementwiseArgs a{x, x_row_stride, residual, res_row_stride, weight, eps, T, H, dtype, out, out_row_stride};
    if (!pq::rms_norm_vectorised(a) && H > pq::RMS_SCALAR_MAX_H)
        return fail(PQ_ERR_UNSUPPORTED, "%s: rows of %d elements need 16-byte aligned tensors, strides and H %% 8 == 0 "
                    "(up to 16384 elements); unaligned rows are supported up to %d", fn, H, pq::RMS_SCALAR_MAX_H);
    hipError_t e = pq::launch_rms_norm(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(e));
    return PQ_OK;
}

int pq_rms_norm(const void *x, int64_t x_row_stride, const void *weight, float eps, int32_t T, int32_t H,
                int32_t dtype, void *out, int64_t out_row_stride, void *stream) {
    return rms_common("pq_rms_norm", x, x_row_stride, nullptr, 0, weight, eps, T, H, dtype, out, out_row_stride, stream);
}

int pq_fused_add_rms_norm(void *x, int64_t x_row_stride, void *residual, int64_t res_row_stride, const void *weight,
                          float eps, int32_t T, int32_t H, int32_t dtype, void *stream) {
    if (T > 0 && H > 0 && !residual) return fail(PQ_ERR_INVALID_ARGUMENT, "pq_fused_add_rms_norm: residual is NULL");
    return rms_common("pq_fused_add_rms_norm", x, x_row_stride, residual, res_row_stride, weight, eps, T, H, dtype, x,
                      x_row_stride, stream);
}

// ---- gate/up GEMM with the SiLU-and-mul fused in ---------------------------------------------------
static int fill_gate_up(pq::GemmArgs &a, const char *fn, const void *x, int32_t T, int64_t x_row_stride,
                        const void *packed_codes, const void *codebooks, const void *scales, const void *bias,
                        int32_t I, int32_t G, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y,
                        int64_t y_row_stride) {
    if (I < 0 || I > INT32_MAX / 2) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: I=%d", fn, I);
    const int32_t parts[2] = {I, I};
    int rc = fill_gemm(a, fn, x, T, x_row_stride, packed_codes, codebooks, scales, bias, parts, 2, 2 * I, G, 1, K, d,
                       code_bytes, dtype, y, 2 * (int64_t)I);
    if (rc) return rc;
    if (y_row_stride < I) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: y_row_stride=%lld < I=%d", fn, (long long)y_row_stride, I);
    a.y_stride = y_row_stride;
    a.packed = true;
    return PQ_OK;
}

int pq_gemm_silu_mul_is_fused(int32_t T, int32_t I, int32_t G, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype) {
    pq::GemmArgs a;
    void *al = reinterpret_cast<void *>(uintptr_t(256));
    if (fill_gate_up(a, "pq_gemm_silu_mul_is_fused", al, T, (int64_t)G * d, al, al, al, nullptr, I, G, K, d, code_bytes,
                     dtype, al, I))
        return 0;
    return pq::packable(a.parts, G, d, K, 1, code_bytes) && pq::gemm_stream_act_supported(a) ? 1 : 0;
}

// The decode kernel's fused launch where it takes the problem (at most 16 tokens here: the entry's fused range) and
// where the decode kernel is the faster one for the unfused layer too (gemm_decode_preferred's rule), else the
// streaming kernel's.
static bool decode_act_taken(const pq::GemmArgs &a) {
    return a.T <= 16 && (a.G / 64 > 2 || a.N > 4096) && pq::gemm_decode_act_supported(a);
}

int pq_gemm_packed_silu_mul(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                            const void *codebooks, const void *scales, const void *bias, int32_t I, int32_t G,
                            int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride,
                            void *tmp, void *workspace, size_t workspace_bytes, void *stream) {
    const char *fn = "pq_gemm_packed_silu_mul";
    pq::GemmArgs a;
    int rc = fill_gate_up(a, fn, x, T, x_row_stride, packed_codes, codebooks, scales, bias, I, G, K, d, code_bytes,
                          dtype, y, y_row_stride);
    if (rc) return rc;
    if (T == 0 || I == 0) return PQ_OK;
    if (!y || !scales || !x || !packed_codes || !codebooks) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    if (!pq::packable(a.parts, G, d, K, 1, code_bytes))
        return fail(PQ_ERR_UNSUPPORTED, "%s: this format has no packed layout (pq_packed_codes_bytes() == 0)", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pq::gemm_stream_act_supported(a)) {
        hipError_t e = decode_act_taken(a) ? pq::launch_gemm_decode_act(a, s) : pq::launch_gemm_stream_act(a, s);
        if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(e));
        return PQ_OK;
    }
    // not fused for this format / batch: the GEMM into tmp [T][2 I], then the stand-alone activation
    if (!tmp)
        return fail(PQ_ERR_WORKSPACE, "%s: T=%d is not fused for this format (pq_gemm_silu_mul_is_fused() == 0): "
                    "tmp [T][2*I] is required", fn, T);
    const int32_t parts[2] = {I, I};
    rc = run_gemm(fn, true, x, T, x_row_stride, packed_codes, codebooks, scales, bias, parts, 2, 2 * I, G, 1, K, d,
                  code_bytes, dtype, tmp, 2 * (int64_t)I, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return pq_silu_and_mul(tmp, 2 * (int64_t)I, T, I, dtype, y, y_row_stride, stream);
}

// ---- fused add + RMSNorm carried by the GEMMs either side of it (decode kernel) --------------------------------
static int norm_common(const char *fn, pq::NormFuse &nf, const void *x, int32_t T, int64_t x_row_stride,
                       const void *packed_codes, const void *codebooks, const void *scales, const void *bias,
                       const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d,
                       int32_t code_bytes, int32_t dtype, void *y, int64_t y_row_stride, void *workspace,
                       size_t workspace_bytes, void *stream, bool query) {
    pq::GemmArgs a;
    void *al = reinterpret_cast<void *>(uintptr_t(256));   // queries: aligned dummy pointers
    int rc = fill_gemm(a, fn, query ? al : x, T, query ? (int64_t)G * d : x_row_stride, query ? al : packed_codes,
                       query ? al : codebooks, query ? al : scales, query ? nullptr : bias, part_sizes, num_parts, N, G, M,
                       K, d, code_bytes, dtype, query ? al : y, query ? (int64_t)N : y_row_stride);
    if (rc) return rc;
    if (T == 0 || N == 0) return query ? 0 : PQ_OK;
    if (!query && (!x || !packed_codes || !codebooks || !scales))
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    a.packed = true;
    a.workspace = query ? al : workspace;
    a.workspace_bytes = query ? pq::WS_COUNTER_BYTES + pq::WS_PARTIAL_BYTES : workspace_bytes;
    a.norm = &nf;
    if (!pq::packable(a.parts, G, d, K, M, code_bytes) || !pq::gemm_decode_norm_supported(a)) {
        if (query) return 0;
        return fail(PQ_ERR_UNSUPPORTED, "%s: not a problem the decode kernel fuses (see %s_supported()), or a tensor is "
                    "misaligned", fn, fn);
    }
    if (query) return 1;
    hipError_t e = pq::launch_gemm_decode(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(e));
    return PQ_OK;
}

int pq_gemm_packed_add_norm_supported(int32_t T, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                                      int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype) {
    void *al = reinterpret_cast<void *>(uintptr_t(256));
    pq::NormFuse nf{1, al, (int64_t)N, al, al, (int64_t)N, static_cast<float *>(al), 0, 0.f};
    const int r = norm_common("pq_gemm_packed_add_norm", nf, nullptr, T, 0, nullptr, nullptr, nullptr, nullptr, part_sizes,
                              num_parts, N, G, M, K, d, code_bytes, dtype, nullptr, 0, nullptr, 0, nullptr, true);
    return r == 1 ? 1 : 0;
}

int pq_gemm_packed_normed_supported(int32_t T, const int32_t *part_sizes, int32_t num_parts, int32_t N, int32_t G,
                                    int32_t M, int32_t K, int32_t d, int32_t code_bytes, int32_t dtype) {
    void *al = reinterpret_cast<void *>(uintptr_t(256));
    pq::NormFuse nf{2, nullptr, 0, nullptr, nullptr, 0, static_cast<float *>(al), G * d, 1e-5f};
    const int r = norm_common("pq_gemm_packed_normed", nf, nullptr, T, 0, nullptr, nullptr, nullptr, nullptr, part_sizes,
                              num_parts, N, G, M, K, d, code_bytes, dtype, nullptr, 0, nullptr, 0, nullptr, true);
    return r == 1 ? 1 : 0;
}

int pq_gemm_packed_add_norm(const void *x, int32_t T, int64_t x_row_stride, const void *packed_codes,
                            const void *codebooks, const void *scales, const void *bias, const int32_t *part_sizes,
                            int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                            int32_t dtype, void *residual, int64_t residual_row_stride, const void *norm_weight, void *xw,
                            int64_t xw_row_stride, float *ssq, void *workspace, size_t workspace_bytes, void *stream) {
    const char *fn = "pq_gemm_packed_add_norm";
    if (T > 0 && N > 0 && (!residual || !norm_weight || !xw || !ssq))
        return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    pq::NormFuse nf{1, residual, residual_row_stride, norm_weight, xw, xw_row_stride, ssq, 0, 0.f};
    // y is not written: the GEMM's output lives on as the new residual
    return norm_common(fn, nf, x, T, x_row_stride, packed_codes, codebooks, scales, bias, part_sizes, num_parts, N, G, M, K,
                       d, code_bytes, dtype, residual, residual_row_stride, workspace, workspace_bytes, stream, false);
}

int pq_gemm_packed_normed(const void *xw, int32_t T, int64_t xw_row_stride, const void *packed_codes,
                          const void *codebooks, const void *scales, const void *bias, const int32_t *part_sizes,
                          int32_t num_parts, int32_t N, int32_t G, int32_t M, int32_t K, int32_t d, int32_t code_bytes,
                          int32_t dtype, const float *ssq, float eps, void *y, int64_t y_row_stride, void *workspace,
                          size_t workspace_bytes, void *stream) {
    const char *fn = "pq_gemm_packed_normed";
    if (T > 0 && N > 0 && (!ssq || !y)) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    pq::NormFuse nf{2, nullptr, 0, nullptr, nullptr, 0, const_cast<float *>(ssq), G * d, eps};
    return norm_common(fn, nf, xw, T, xw_row_stride, packed_codes, codebooks, scales, bias, part_sizes, num_parts, N, G, M,
                       K, d, code_bytes, dtype, y, y_row_stride, workspace, workspace_bytes, stream, false);
}

int pq_gemm_packed_normed_silu_mul_supported(int32_t T, int32_t I, int32_t G, int32_t K, int32_t d, int32_t code_bytes,
                                             int32_t dtype) {
    pq::GemmArgs a;
    void *al = reinterpret_cast<void *>(uintptr_t(256));
    if (fill_gate_up(a, "pq_gemm_packed_normed_silu_mul", al, T, (int64_t)G * d, al, al, al, nullptr, I, G, K, d,
                     code_bytes, dtype, al, I))
        return 0;
    if (T == 0 || I == 0) return 0;
    pq::NormFuse nf{2, nullptr, 0, nullptr, nullptr, 0, static_cast<float *>(al), G * d, 1e-5f};
    a.norm = &nf;
    return pq::packable(a.parts, G, d, K, 1, code_bytes) && pq::gemm_decode_act_supported(a) ? 1 : 0;
}

int pq_gemm_packed_normed_silu_mul(const void *xw, int32_t T, int64_t xw_row_stride, const void *packed_codes,
                                   const void *codebooks, const void *scales, const void *bias, int32_t I, int32_t G,
                                   int32_t K, int32_t d, int32_t code_bytes, int32_t dtype, const float *ssq, float eps,
                                   void *y, int64_t y_row_stride, void *stream) {
    const char *fn = "pq_gemm_packed_normed_silu_mul";
    pq::GemmArgs a;
    int rc = fill_gate_up(a, fn, xw, T, xw_row_stride, packed_codes, codebooks, scales, bias, I, G, K, d, code_bytes,
                          dtype, y, y_row_stride);
    if (rc) return rc;
    if (T == 0 || I == 0) return PQ_OK;
    if (!y || !scales || !xw || !packed_codes || !codebooks || !ssq) return fail(PQ_ERR_INVALID_ARGUMENT, "%s: NULL tensor pointer", fn);
    pq::NormFuse nf{2, nullptr, 0, nullptr, nullptr, 0, const_cast<float *>(ssq), G * d, eps};
    a.norm = &nf;
    if (!pq::packable(a.parts, G, d, K, 1, code_bytes) || !pq::gemm_decode_act_supported(a))
        return fail(PQ_ERR_UNSUPPORTED, "%s: not a problem the decode kernel fuses (see %s_supported())", fn, fn);
    hipError_t e = pq::launch_gemm_decode_act(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PQ_ERR_LAUNCH, "%s: %s", fn, hipGetErrorString(e));
    return PQ_OK;
}

}  // extern "C"
