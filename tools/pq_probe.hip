// Measurement aid for bench.py (not part of libpq_amd.so): the rate a bare loop of v_mfma_f32_16x16x32_bf16 reaches on
// this device with the caller's operands in registers -- 2 waves per SIMD on every CU, an 8 x 4 tile of 16x16 fragments
// per wave (the tiled kernel's 512 x 128 shape), no memory traffic.  Under MFMA load the chip holds a clock that depends
// on the operands' bit patterns, so this, with operands drawn like the benchmark's, is the rate a perfect kernel could
// reach there; the nominal peak assumes the top clock.
//   hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 tools/pq_probe.hip -o tools/libpq_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// `in`: n_in uint4 (8 bf16 each), n_in a power of two; every lane picks 12 of them
__global__ __launch_bounds__(512) void probe_mfma_kernel(const uint4 *__restrict__ in, uint32_t mask, float *__restrict__ out, int iters) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(gid * 12 + i) & mask]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(gid * 12 + 8 + i) & mask]);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * pp][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * pp], b[j], acc[2 * pp][j], 0, 0, 0);
                acc[2 * pp + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * pp + 1], b[j], acc[2 * pp + 1][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    out[gid] = s;
}

extern "C" {
// One launch of `blocks` x 512 threads; FLOPs = blocks * 8 waves * iters * 32 MFMAs * 16384.  `out` holds blocks * 512 floats.
int pq_probe_mfma(const void *in, int n_in, void *out, int blocks, int iters, void *stream) {
    if (!in || !out || n_in < 16 || (n_in & (n_in - 1)) || blocks < 1 || iters < 1) return 1;
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(in), (uint32_t)n_in - 1u, static_cast<float *>(out), iters);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}

// Cache probe: block b reads one dword of every 128-byte line of bytes [b * bytes_per_block, (b + 1) * bytes_per_block)
// of `base` -- the lines reach the L2 of the XCD the block runs on (block b -> XCD b % 8) while 4 bytes of each cross
// into the CU.  `shift` moves the blocks' regions by that many blocks (a deliberately wrong XCD).
__global__ __launch_bounds__(64) void probe_touch_kernel(const unsigned char *__restrict__ base, size_t bytes_per_block,
                                                          size_t total, int shift, uint32_t *__restrict__ sink) {
    const size_t blk = (blockIdx.x + (size_t)shift) % gridDim.x;
    const size_t off0 = blk * bytes_per_block;
    uint32_t acc = 0;
    for (size_t o = (size_t)threadIdx.x * 128; o < bytes_per_block; o += (size_t)blockDim.x * 128)
        if (off0 + o + 4 <= total) acc += *reinterpret_cast<const uint32_t *>(base + off0 + o);
    if (acc == 0x12345678u) sink[0] = acc;   // keeps the loads
}

extern "C" int pq_probe_touch(const void *base, size_t bytes_per_block, size_t total, int blocks, int shift, void *sink,
                              void *stream) {
    if (!base || !sink || blocks < 1) return 1;
    hipLaunchKernelGGL(probe_touch_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char *>(base), bytes_per_block, total, shift, static_cast<uint32_t *>(sink));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
