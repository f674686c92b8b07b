// Small-message all-reduce (SUM) over peer-mapped device buffers: what a row-parallel PQ linear needs at decode
// batch sizes, where a [T][hidden] partial result is a few KiB to a few MiB and a ring collective is all latency.
// One process per GPU; every rank owns one buffer (pq_ar_alloc) which the others map through its IPC handle
// (pq_ar_open); xGMI carries plain loads and stores between them.  It takes the place of the reference's custom
// all-reduce for this path (csrc/custom_all_reduce.cuh:214-280 kernels, :478-491 dispatch by size,
// vllm/distributed/device_communicators/custom_all_reduce.py).
//
// Buffer of a rank (all ranks use the same layout):
//   [0, 4 KiB)        private words: status, then one call counter per lane
//   flags1, flags2    [world][AR_LANES] words each, written by the PEERS (slot [writer's rank][lane])
//   data[2][AR_LANES][AR_LANE_BYTES]   this rank's input, staged for the peers to read
//   red [2][AR_LANES][AR_LANE_BYTES]   two-shot only: the sub-slices this rank has reduced
// A launch cuts the message into at most AR_LANES chunks, one block ("lane") each.  A lane is a communicator of its
// own: its counter e (the number of calls the lane has taken part in, identical on all ranks because they issue the
// same calls) selects the half e & 1 of its data / red areas and is the value of its flags.
//   one shot (small messages):  stage my chunk; flag the peers; wait for their flags; sum all ranks' chunks in rank
//                               order (the same order everywhere: bit-identical results on all ranks)
//   two shots:                  stage; flag; wait; sum sub-slice `rank` of the chunk over all ranks into red and out;
//                               flag 2; wait 2; copy the other ranks' sub-slices from their red areas
// No barrier at the end: a lane overwrites half e & 1 again in its call e + 2, which it starts only after its call
// e + 1 has finished, i.e. after every peer has flagged call e + 1 of the lane, which a peer does only after its
// own kernel of call e has completed (stream order).
// The buffers are allocated uncached (fine-grained), flags are released / acquired at system scope, and a wait gives
// up after the communicator's timeout (30 s unless pq_ar_set_timeout_ms changed it; 100 MHz wall clock), so that
// every wave ends.  A call whose wait timed out has no sum to give: it fills its part of every output with NaN (all
// bits set) and sets the buffer's status word, which stays set (pq_ar_status).  The caller must treat that as fatal
// for the communicator: the lane counters of the ranks may have come apart.
//
// all_reduce_add_norm: the same one-shot sum with the fused residual add + RMSNorm that follows a row-parallel
// layer in a decoder block (vllm/model_executor/models/llama.py:272-284) in its epilogue.  Lanes take whole rows
// of the [T][H] message; a block of H / 8 threads (or 1024 threads with two pieces each) holds a row in registers
// exactly as rms_norm_kernel does (pq_elementwise.hip), so the results are those of the all-reduce followed by
// pq_fused_add_rms_norm bit for bit: sum over the ranks rounded to the act dtype, + residual in fp32, the rounded
// sum is the new residual, the unrounded one feeds the mean of squares, z * rsqrt(var + eps) rounded, * weight
// rounded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "pq_common.cuh"

namespace pq {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_LANES = 32;
constexpr size_t AR_LANE_BYTES = 65536;
constexpr size_t AR_PRIVATE_BYTES = 4096;
constexpr size_t AR_FLAG_BYTES = (size_t)AR_MAX_RANKS * AR_LANES * 4;
constexpr size_t AR_OFF_FLAGS1 = AR_PRIVATE_BYTES;
constexpr size_t AR_OFF_FLAGS2 = AR_OFF_FLAGS1 + AR_FLAG_BYTES;
constexpr size_t AR_OFF_DATA = 8192;
constexpr size_t AR_HALF_BYTES = (size_t)AR_LANES * AR_LANE_BYTES;
constexpr size_t AR_OFF_RED = AR_OFF_DATA + 2 * AR_HALF_BYTES;
constexpr size_t AR_BUFFER_BYTES = AR_OFF_RED + 2 * AR_HALF_BYTES;
constexpr size_t AR_ONE_SHOT_MAX = 262144;                       // above: two shots
constexpr size_t AR_MAX_MESSAGE = AR_HALF_BYTES;                 // 2 MiB
constexpr unsigned long long AR_TIMEOUT_TICKS = 3000000000ull;   // 30 s: far above any skew between ranks of one step
constexpr size_t AR_NORM_MAX = 524288;                           // largest message of the fused add + norm variant
static_assert(AR_OFF_FLAGS2 + AR_FLAG_BYTES <= AR_OFF_DATA, "flags overlap the data");

struct ArComm {
    int rank, world;
    unsigned char *buf[AR_MAX_RANKS];     // buf[rank] = own buffer, the others as mapped into this process
    unsigned long long timeout_ticks;
};

struct ArParams {
    unsigned char *buf[AR_MAX_RANKS];
    const unsigned char *in;
    unsigned char *out;
    uint32_t bytes;          // of the message (a multiple of 16)
    uint32_t chunk;          // bytes per lane (a multiple of 16 * world for two shots)
    int rank, world, dtype, two_shot;
    unsigned long long timeout_ticks;
    // fused add + RMSNorm (all_reduce_add_norm only): the message is [T][H]; out is not written
    uint16_t *residual;      // [T][H], read and replaced
    const uint16_t *norm_w;  // [H]
    uint16_t *xw;            // [T][H] normalised output
    int H, rows_per_lane;
    float eps;
};

__device__ inline void ar_flag_peers(const ArParams &p, size_t flags_off, int lane, uint32_t e) {
    // all of the block's stores are fenced (every thread fenced before the barrier in front of this)
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        uint32_t *f = reinterpret_cast<uint32_t *>(p.buf[t] + flags_off) + p.rank * AR_LANES + lane;
        __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Every peer has flagged call e of the lane: true for every thread of the block.  False (for every thread) when a
// wait timed out: the buffer's status word is set and *timed_out (LDS, zero at kernel start) stays set.
__device__ inline bool ar_wait_peers(const ArParams &p, size_t flags_off, int lane, uint32_t e, int *timed_out) {
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        const uint32_t *f = reinterpret_cast<const uint32_t *>(p.buf[p.rank] + flags_off) + t * AR_LANES + lane;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // counters only grow; a peer that is a call ahead has flagged this one as well
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                __hip_atomic_store(reinterpret_cast<uint32_t *>(p.buf[p.rank]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    return __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
}

// a call without a sum: NaN (all bits set: a NaN of fp16, bf16 and fp32 alike) over `n` bytes at dst
__device__ inline void ar_poison(unsigned char *dst, uint32_t n) {
    const u32x4 nan{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    for (uint32_t o = threadIdx.x * 16u; o < n; o += blockDim.x * 16u) *reinterpret_cast<u32x4 *>(dst + o) = nan;
}

template <int DT>   // 0 fp16, 1 bf16, 2 fp32: sum of 16 bytes from every rank, in rank order, fp32 accumulation
__device__ inline u32x4 ar_sum16(const ArParams &p, size_t off) {
    float acc[8];
    u32x4 v = *reinterpret_cast<const u32x4 *>(p.buf[0] + off);
    if constexpr (DT == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t w = v[i]; acc[i] = __uint_as_float(w); }   // (no bit_cast of a vector element)
    } else {
        typedef typename std::conditional<DT == 1, bf16_t, fp16_t>::type AT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = Act<AT>::to_f32((uint16_t)(v[i] & 0xffffu));
            acc[2 * i + 1] = Act<AT>::to_f32((uint16_t)(v[i] >> 16));
        }
    }
    for (int r = 1; r < p.world; ++r) {
        v = *reinterpret_cast<const u32x4 *>(p.buf[r] + off);
        if constexpr (DT == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t w = v[i]; acc[i] += __uint_as_float(w); }
        } else {
            typedef typename std::conditional<DT == 1, bf16_t, fp16_t>::type AT;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] += Act<AT>::to_f32((uint16_t)(v[i] & 0xffffu));
                acc[2 * i + 1] += Act<AT>::to_f32((uint16_t)(v[i] >> 16));
            }
        }
    }
    u32x4 o;
    if constexpr (DT == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(acc[i]);
    } else {
        typedef typename std::conditional<DT == 1, bf16_t, fp16_t>::type AT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (uint32_t)Act<AT>::from_f32(acc[2 * i]) | ((uint32_t)Act<AT>::from_f32(acc[2 * i + 1]) << 16);
    }
    return o;
}

template <int DT>
__global__ __launch_bounds__(256) void all_reduce_kernel(const ArParams p) {
    __shared__ int s_timed_out;
    const int lane = blockIdx.x, t = threadIdx.x;
    unsigned char *own = p.buf[p.rank];
    uint32_t *counter = reinterpret_cast<uint32_t *>(own) + 1 + lane;
    const uint32_t e = *counter + 1u;                      // this call of the lane (read by every thread before the
    if (t == 0) s_timed_out = 0;
    __syncthreads();                                       // write at the end)
    const size_t half = (size_t)(e & 1u) * AR_HALF_BYTES + (size_t)lane * AR_LANE_BYTES;
    const size_t c0 = (size_t)lane * p.chunk;              // the lane's chunk of the message
    const uint32_t cb = p.bytes - c0 < p.chunk ? (uint32_t)(p.bytes - c0) : p.chunk;

    // stage the chunk for the peers
    for (uint32_t o = t * 16u; o < cb; o += 256u * 16u)
        *reinterpret_cast<u32x4 *>(own + AR_OFF_DATA + half + o) = *reinterpret_cast<const u32x4 *>(p.in + c0 + o);
    __threadfence_system();
    __syncthreads();
    ar_flag_peers(p, AR_OFF_FLAGS1, lane, e);
    if (!ar_wait_peers(p, AR_OFF_FLAGS1, lane, e, &s_timed_out)) {
        // no sum to give; the peers still get this call's second flag, so that they do not wait for it as well
        ar_poison(p.out + c0, cb);
        if (p.two_shot) {
            __threadfence_system();
            __syncthreads();
            ar_flag_peers(p, AR_OFF_FLAGS2, lane, e);
        }
        if (t == 0) *counter = e;
        return;
    }

    if (!p.two_shot) {
        for (uint32_t o = t * 16u; o < cb; o += 256u * 16u)
            *reinterpret_cast<u32x4 *>(p.out + c0 + o) = ar_sum16<DT>(p, AR_OFF_DATA + half + o);
    } else {
        // sub-slice r of the chunk: 16-byte units split evenly, the remainder to the last rank
        const uint32_t units = cb / 16u, per = units / p.world;
        const uint32_t s0 = (uint32_t)p.rank * per * 16u;
        const uint32_t s1 = p.rank == p.world - 1 ? cb : s0 + per * 16u;
        for (uint32_t o = s0 + t * 16u; o < s1; o += 256u * 16u) {
            const u32x4 v = ar_sum16<DT>(p, AR_OFF_DATA + half + o);
            *reinterpret_cast<u32x4 *>(own + AR_OFF_RED + half + o) = v;
            *reinterpret_cast<u32x4 *>(p.out + c0 + o) = v;
        }
        __threadfence_system();
        __syncthreads();
        ar_flag_peers(p, AR_OFF_FLAGS2, lane, e);
        if (!ar_wait_peers(p, AR_OFF_FLAGS2, lane, e, &s_timed_out)) {
            ar_poison(p.out + c0, cb);
            if (t == 0) *counter = e;
            return;
        }
        for (int r = 0; r < p.world; ++r) {
            if (r == p.rank) continue;
            const uint32_t r0 = (uint32_t)r * per * 16u;
            const uint32_t r1 = r == p.world - 1 ? cb : r0 + per * 16u;
            for (uint32_t o = r0 + t * 16u; o < r1; o += 256u * 16u)
                *reinterpret_cast<u32x4 *>(p.out + c0 + o) = *reinterpret_cast<const u32x4 *>(p.buf[r] + AR_OFF_RED + half + o);
        }
    }
    if (t == 0) *counter = e;
}

// sum over the block in rms_norm_kernel's order (lane tree, then the waves in order): the same value in every thread
template <int THREADS>
__device__ inline float ar_block_sum(float v, float *s_part) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) tot += s_part[w];
    __syncthreads();                                       // s_part is written again for the next row
    return tot;
}

// One-shot sum + residual add + RMSNorm.  THREADS x NV pieces of 8 elements cover a row; lane b takes rows
// [b * rows_per_lane, ...) of the message.
template <typename AT, int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void all_reduce_add_norm_kernel(const ArParams p) {
    constexpr int DT = std::is_same<AT, bf16_t>::value ? 1 : 0;
    __shared__ int s_timed_out;
    __shared__ float s_part[THREADS / 64];
    const int lane = blockIdx.x, t = threadIdx.x;
    unsigned char *own = p.buf[p.rank];
    uint32_t *counter = reinterpret_cast<uint32_t *>(own) + 1 + lane;
    const uint32_t e = *counter + 1u;
    if (t == 0) s_timed_out = 0;
    __syncthreads();
    const size_t half = (size_t)(e & 1u) * AR_HALF_BYTES + (size_t)lane * AR_LANE_BYTES;
    const size_t c0 = (size_t)lane * p.chunk;
    const uint32_t cb = p.bytes - c0 < p.chunk ? (uint32_t)(p.bytes - c0) : p.chunk;
    const uint32_t row_bytes = (uint32_t)p.H * 2u;
    const int rows = (int)(cb / row_bytes), row0 = lane * p.rows_per_lane;

    for (uint32_t o = t * 16u; o < cb; o += THREADS * 16u)
        *reinterpret_cast<u32x4 *>(own + AR_OFF_DATA + half + o) = *reinterpret_cast<const u32x4 *>(p.in + c0 + o);
    __threadfence_system();
    __syncthreads();
    ar_flag_peers(p, AR_OFF_FLAGS1, lane, e);
    if (!ar_wait_peers(p, AR_OFF_FLAGS1, lane, e, &s_timed_out)) {
        ar_poison(reinterpret_cast<unsigned char *>(p.residual) + c0, cb);
        ar_poison(reinterpret_cast<unsigned char *>(p.xw) + c0, cb);
        if (t == 0) *counter = e;
        return;
    }
    const int nvec = p.H / 8;
    for (int r = 0; r < rows; ++r) {
        uint16_t *rr = p.residual + (size_t)(row0 + r) * p.H;
        float z[NV][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = i * THREADS + t;
            if (v < nvec) {
                // the all-reduce's output as the unfused path has it: the sum over the ranks rounded to the act dtype
                const u32x4 a = ar_sum16<DT>(p, AR_OFF_DATA + half + (size_t)r * row_bytes + (size_t)v * 16u);
                const u32x4 rs_ = *reinterpret_cast<const u32x4 *>(rr + v * 8);
                u32x4 zr;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float f = Act<AT>::to_f32((uint16_t)(a[w] >> (16 * h)));
                        f += Act<AT>::to_f32((uint16_t)(rs_[w] >> (16 * h)));
                        packed |= (uint32_t)Act<AT>::from_f32(f) << (16 * h);
                        z[i][2 * w + h] = f;
                        ss += f * f;
                    }
                    zr[w] = packed;
                }
                *reinterpret_cast<u32x4 *>(rr + v * 8) = zr;
            }
        }
        const float rs = rsqrtf(ar_block_sum<THREADS>(ss, s_part) / (float)p.H + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = i * THREADS + t;
            if (v < nvec) {
                const u32x4 wv = *reinterpret_cast<const u32x4 *>(p.norm_w + v * 8);
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float n = Act<AT>::to_f32(Act<AT>::from_f32(rounded_f32(z[i][2 * w + h] * rs)));
                        packed |= (uint32_t)Act<AT>::from_f32(n * Act<AT>::to_f32((uint16_t)(wv[w] >> (16 * h)))) << (16 * h);
                    }
                    o[w] = packed;
                }
                *reinterpret_cast<u32x4 *>(p.xw + (size_t)(row0 + r) * p.H + v * 8) = o;
            }
        }
    }
    if (t == 0) *counter = e;
}

}  // namespace pq

// ---- C ABI (include/pq_amd.h) -----------------------------------------------------------------------------------
namespace {
thread_local char g_ar_err[256] = "";
int ar_fail(int code, const char *what, hipError_t e) {
    snprintf(g_ar_err, sizeof(g_ar_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}
}  // namespace

extern "C" {

const char *pq_ar_last_error(void) { return g_ar_err; }

size_t pq_ar_buffer_bytes(void) { return pq::AR_BUFFER_BYTES; }
size_t pq_ar_max_message_bytes(void) { return pq::AR_MAX_MESSAGE; }

int pq_ar_alloc(void **buffer, void *ipc_handle) {
    if (!buffer || !ipc_handle) return ar_fail(1, "pq_ar_alloc: NULL argument", hipSuccess);
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, pq::AR_BUFFER_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) return ar_fail(4, "pq_ar_alloc: hipExtMallocWithFlags", e);
    e = hipMemset(p, 0, pq::AR_OFF_DATA);                  // status, counters, flags
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return ar_fail(4, "pq_ar_alloc: hipIpcGetMemHandle", e); }
    static_assert(sizeof(h) == 64, "IPC handle size");
    memcpy(ipc_handle, &h, sizeof(h));
    *buffer = p;
    return 0;
}

int pq_ar_free(void *buffer) {
    hipError_t e = hipFree(buffer);
    return e == hipSuccess ? 0 : ar_fail(4, "pq_ar_free", e);
}

int pq_ar_open(const void *ipc_handle, void **peer_buffer) {
    if (!ipc_handle || !peer_buffer) return ar_fail(1, "pq_ar_open: NULL argument", hipSuccess);
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(peer_buffer, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : ar_fail(4, "pq_ar_open: hipIpcOpenMemHandle", e);
}

int pq_ar_close(void *peer_buffer) {
    hipError_t e = hipIpcCloseMemHandle(peer_buffer);
    return e == hipSuccess ? 0 : ar_fail(4, "pq_ar_close", e);
}

int pq_ar_create(int32_t rank, int32_t world, void *const *buffers, void **comm) {
    if (!buffers || !comm || world < 2 || world > pq::AR_MAX_RANKS || rank < 0 || rank >= world)
        return ar_fail(1, "pq_ar_create: 2..8 ranks, rank inside, buffers of every rank", hipSuccess);
    pq::ArComm *c = static_cast<pq::ArComm *>(calloc(1, sizeof(pq::ArComm)));
    if (!c) return ar_fail(4, "pq_ar_create: out of host memory", hipSuccess);
    c->rank = rank;
    c->world = world;
    c->timeout_ticks = pq::AR_TIMEOUT_TICKS;
    for (int r = 0; r < world; ++r) {
        if (!buffers[r]) { free(c); return ar_fail(1, "pq_ar_create: NULL buffer", hipSuccess); }
        c->buf[r] = static_cast<unsigned char *>(buffers[r]);
    }
    *comm = c;
    return 0;
}

int pq_ar_destroy(void *comm) {
    free(comm);
    return 0;
}

int pq_ar_set_timeout_ms(void *comm, int64_t ms) {
    pq::ArComm *c = static_cast<pq::ArComm *>(comm);
    if (!c || ms <= 0) return ar_fail(1, "pq_ar_set_timeout_ms: a communicator and a positive time", hipSuccess);
    c->timeout_ticks = (unsigned long long)ms * 100000ull;   // 100 MHz wall clock
    return 0;
}

int pq_ar_all_reduce(void *comm, const void *in, void *out, int64_t count, int32_t dtype, void *stream) {
    const pq::ArComm *c = static_cast<const pq::ArComm *>(comm);
    if (!c || !in || !out) return ar_fail(1, "pq_ar_all_reduce: NULL argument", hipSuccess);
    if (dtype < 0 || dtype > 2) return ar_fail(1, "pq_ar_all_reduce: dtype is fp16 (0), bf16 (1) or fp32 (2)", hipSuccess);
    if (count <= 0) return count == 0 ? 0 : ar_fail(1, "pq_ar_all_reduce: negative count", hipSuccess);
    const int64_t bytes = count * (dtype == 2 ? 4 : 2);
    if (bytes % 16 || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15))
        return ar_fail(2, "pq_ar_all_reduce: the message must be a multiple of 16 bytes, in and out 16-byte aligned", hipSuccess);
    if ((size_t)bytes > pq::AR_MAX_MESSAGE)
        return ar_fail(2, "pq_ar_all_reduce: message above pq_ar_max_message_bytes()", hipSuccess);
    pq::ArParams p;
    for (int r = 0; r < pq::AR_MAX_RANKS; ++r) p.buf[r] = r < c->world ? c->buf[r] : nullptr;
    p.in = static_cast<const unsigned char *>(in);
    p.out = static_cast<unsigned char *>(out);
    p.bytes = (uint32_t)bytes;
    p.rank = c->rank;
    p.world = c->world;
    p.dtype = dtype;
    p.two_shot = (size_t)bytes > pq::AR_ONE_SHOT_MAX ? 1 : 0;
    p.timeout_ticks = c->timeout_ticks;
    p.residual = nullptr; p.norm_w = nullptr; p.xw = nullptr; p.H = 0; p.rows_per_lane = 0; p.eps = 0.f;
    // lanes: chunks of at least 4 KiB (a block of 256 threads moves 4 KiB per pass), a multiple of 16 * world bytes
    const uint32_t gran = 16u * (uint32_t)c->world;
    uint32_t chunk = (uint32_t)((bytes + pq::AR_LANES - 1) / pq::AR_LANES);
    if (chunk < 4096u) chunk = 4096u;
    chunk = (chunk + gran - 1) / gran * gran;
    if (chunk > pq::AR_LANE_BYTES) return ar_fail(2, "pq_ar_all_reduce: chunk above a lane's capacity", hipSuccess);
    p.chunk = chunk;
    const int lanes = (int)((bytes + chunk - 1) / chunk);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0) hipLaunchKernelGGL(pq::all_reduce_kernel<0>, dim3(lanes), dim3(256), 0, s, p);
    else if (dtype == 1) hipLaunchKernelGGL(pq::all_reduce_kernel<1>, dim3(lanes), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pq::all_reduce_kernel<2>, dim3(lanes), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ar_fail(4, "pq_ar_all_reduce: launch", e);
}

// rows per lane of the fused add + norm variant for a [T][H] message, 0 when it does not take the problem
static int ar_norm_rows_per_lane(int32_t T, int32_t H, int32_t dtype) {
    if (T <= 0 || H <= 0 || H % 8 || H > 16384 || (dtype != PQ_DTYPE_FP16 && dtype != PQ_DTYPE_BF16)) return 0;
    const size_t row = (size_t)H * 2;
    if ((size_t)T * row > pq::AR_NORM_MAX) return 0;
    const int rpl = (T + pq::AR_LANES - 1) / pq::AR_LANES;
    return (size_t)rpl * row <= pq::AR_LANE_BYTES ? rpl : 0;
}

int pq_ar_add_norm_supported(int32_t T, int32_t H, int32_t dtype) { return ar_norm_rows_per_lane(T, H, dtype) > 0 ? 1 : 0; }

int pq_ar_all_reduce_add_norm(void *comm, const void *in, int32_t T, int32_t H, int32_t dtype, void *residual,
                              const void *norm_weight, float eps, void *xw, void *stream) {
    const char *fn = "pq_ar_all_reduce_add_norm";
    const pq::ArComm *c = static_cast<const pq::ArComm *>(comm);
    if (!c) return ar_fail(1, "pq_ar_all_reduce_add_norm: NULL communicator", hipSuccess);
    if (T == 0 || H == 0) return 0;
    if (!in || !residual || !norm_weight || !xw) return ar_fail(1, "pq_ar_all_reduce_add_norm: NULL tensor pointer", hipSuccess);
    const int rpl = ar_norm_rows_per_lane(T, H, dtype);
    if (rpl == 0)
        return ar_fail(2, "pq_ar_all_reduce_add_norm: fp16 / bf16 rows of a multiple of 8 up to 16384 elements, at most "
                          "512 KiB in all (pq_ar_add_norm_supported)", hipSuccess);
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(norm_weight) |
         reinterpret_cast<uintptr_t>(xw)) & 15)
        return ar_fail(2, "pq_ar_all_reduce_add_norm: tensors must be 16-byte aligned", hipSuccess);
    (void)fn;
    pq::ArParams p;
    for (int r = 0; r < pq::AR_MAX_RANKS; ++r) p.buf[r] = r < c->world ? c->buf[r] : nullptr;
    p.in = static_cast<const unsigned char *>(in);
    p.out = nullptr;
    p.bytes = (uint32_t)((size_t)T * H * 2);
    p.chunk = (uint32_t)((size_t)rpl * H * 2);
    p.rank = c->rank; p.world = c->world; p.dtype = dtype; p.two_shot = 0;
    p.timeout_ticks = c->timeout_ticks;
    p.residual = static_cast<uint16_t *>(residual);
    p.norm_w = static_cast<const uint16_t *>(norm_weight);
    p.xw = static_cast<uint16_t *>(xw);
    p.H = H; p.rows_per_lane = rpl; p.eps = eps;
    const int lanes = (T + rpl - 1) / rpl;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nvec = H / 8;
    // the block shapes of rms_norm_kernel for the same row length: the same order of the sum of squares
#define PQ_ARN(AT, THREADS, NV) hipLaunchKernelGGL((pq::all_reduce_add_norm_kernel<AT, THREADS, NV>), dim3(lanes), dim3(THREADS), 0, s, p)
#define PQ_ARN_AT(AT)                                  \
    do {                                               \
        if (nvec <= 128) PQ_ARN(AT, 128, 1);           \
        else if (nvec <= 256) PQ_ARN(AT, 256, 1);      \
        else if (nvec <= 512) PQ_ARN(AT, 512, 1);      \
        else if (nvec <= 1024) PQ_ARN(AT, 1024, 1);    \
        else PQ_ARN(AT, 1024, 2);                      \
    } while (0)
    if (dtype == PQ_DTYPE_BF16) PQ_ARN_AT(pq::bf16_t); else PQ_ARN_AT(pq::fp16_t);
#undef PQ_ARN_AT
#undef PQ_ARN
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ar_fail(4, "pq_ar_all_reduce_add_norm: launch", e);
}

int pq_ar_status(void *comm) {
    const pq::ArComm *c = static_cast<const pq::ArComm *>(comm);
    if (!c) return ar_fail(1, "pq_ar_status: NULL", hipSuccess);
    uint32_t st = 0;
    hipError_t e = hipMemcpy(&st, c->buf[c->rank], 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return ar_fail(4, "pq_ar_status", e);
    return st == 0 ? 0 : ar_fail(5, "pq_ar_all_reduce: a wait for a peer's flag timed out (results of that call are invalid)", hipSuccess);
}

}  // extern "C"
