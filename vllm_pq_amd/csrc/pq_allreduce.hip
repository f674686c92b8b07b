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
// up after AR_TIMEOUT_TICKS of the 100 MHz wall clock (status word set, results invalid): every wave ends.
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
constexpr unsigned long long AR_TIMEOUT_TICKS = 200000000ull;    // 2 s
static_assert(AR_OFF_FLAGS2 + AR_FLAG_BYTES <= AR_OFF_DATA, "flags overlap the data");

struct ArComm {
    int rank, world;
    unsigned char *buf[AR_MAX_RANKS];     // buf[rank] = own buffer, the others as mapped into this process
};

struct ArParams {
    unsigned char *buf[AR_MAX_RANKS];
    const unsigned char *in;
    unsigned char *out;
    uint32_t bytes;          // of the message (a multiple of 16)
    uint32_t chunk;          // bytes per lane (a multiple of 16 * world for two shots)
    int rank, world, dtype, two_shot;
};

__device__ inline void ar_flag_peers(const ArParams &p, size_t flags_off, int lane, uint32_t e) {
    // all of the block's stores are fenced (every thread fenced before the barrier in front of this)
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        uint32_t *f = reinterpret_cast<uint32_t *>(p.buf[t] + flags_off) + p.rank * AR_LANES + lane;
        __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// every peer has flagged call e of the lane (or the wait timed out: status word set, false)
__device__ inline void ar_wait_peers(const ArParams &p, size_t flags_off, int lane, uint32_t e) {
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        const uint32_t *f = reinterpret_cast<const uint32_t *>(p.buf[p.rank] + flags_off) + t * AR_LANES + lane;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // counters only grow; a peer that is a call ahead has flagged this one as well
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > AR_TIMEOUT_TICKS) {
                __hip_atomic_store(reinterpret_cast<uint32_t *>(p.buf[p.rank]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
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
    const int lane = blockIdx.x, t = threadIdx.x;
    unsigned char *own = p.buf[p.rank];
    uint32_t *counter = reinterpret_cast<uint32_t *>(own) + 1 + lane;
    const uint32_t e = *counter + 1u;                      // this call of the lane (read by every thread before the
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
    ar_wait_peers(p, AR_OFF_FLAGS1, lane, e);

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
        ar_wait_peers(p, AR_OFF_FLAGS2, lane, e);
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

int pq_ar_status(void *comm) {
    const pq::ArComm *c = static_cast<const pq::ArComm *>(comm);
    if (!c) return ar_fail(1, "pq_ar_status: NULL", hipSuccess);
    uint32_t st = 0;
    hipError_t e = hipMemcpy(&st, c->buf[c->rank], 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return ar_fail(4, "pq_ar_status", e);
    return st == 0 ? 0 : ar_fail(5, "pq_ar_all_reduce: a wait for a peer's flag timed out (results of that call are invalid)", hipSuccess);
}

}  // extern "C"
