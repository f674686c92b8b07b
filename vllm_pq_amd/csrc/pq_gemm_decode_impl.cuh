// Fused codebook gather + MFMA GEMM for decode batches on gfx950 (16 tokens per block; batches of up to 128 tokens run
// as groups of 16 over gridDim.z): d = 8, one codebook of at most 256 entries, one-byte codes in the PACKED order
// (pq_gemm_stream.hip).
//
// y[t][n] = scale[n] * sum_k x[t][k] * W[n][k] + bias[n],   W[n][8g..8g+7] = centroid(code[n][g])
//
// Same arithmetic as the streaming kernel (pq_gemm_stream_impl.cuh): the weight fragment of
// v_mfma_f32_16x16x32 is read straight out of the replicated LDS codebook image at the address the lane's code
// byte selects, the activations are the B operand and stay in registers, K is split over the waves of a block
// and the partial sums meet in LDS in a fixed order.  What differs is how the launch is built, because a decode
// launch lasts a few microseconds and three things set its length that do not matter in a long kernel:
//
//  * Instruction fetch.  Every launch starts with a cold instruction cache, and straight-line code that is
//    executed once runs at the rate the cache fills (measured: 12.5 cycles per 8-byte instruction cold against
//    5 warm with 8 waves per CU), so 1 KiB of code that is executed once costs about 0.7 us.  This kernel is
//    a few short loops: a unit (one 16-row fragment x one 512-feature chunk: 16 gathers + 16 MFMAs) is ONE loop
//    body, whatever the tile shape; tile shape, waves per block and K split are run-time values.
//  * The shape of the activation loads.  Every block reads its K range of all 16 tokens; the memory pipeline takes a
//    wave-load whose quads of lanes each read 64 contiguous bytes four times faster than one shaped like the MFMA
//    operand (four rows per quad).  The activations come in as one contiguous KiB per token by LDS-DMA and are read
//    into the operand registers from LDS (see the kernel body).
//  * The order of the requests.  A CU returns memory data in request order across all its waves, so the code
//    pieces (HBM) are requested only when every wave's activations (L2) have landed; they never pass through
//    registers: every wave sends its units' pieces (1 KiB each) straight into an LDS staging area by LDS-DMA, and a
//    unit is computed when its piece has landed (the wait is a counted vmcnt).  A unit's accumulator lives in the
//    staging slot its codes came in, so the loop body does not depend on the fragment either.
//
// Work split: a block of WK x WR waves owns WR * nf fragments (16 rows each) of one output partition and at most 8
// chunks of 512 features; wave (wk, wr) takes chunk wk of fragments wr * nf .. + nf - 1.  A longer K is split over
// gridDim.y blocks per tile, whose fp32 partial sums meet in the caller's workspace (the streaming kernel's
// protocol).  The activations of the chunk (16 x 16 bytes per lane) are read once and reused by the nf fragments.
#pragma once
#include <type_traits>
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

constexpr int DC_MAX_WAVES = 8;
constexpr int DC_MAX_TOKENS = 128;                // batches beyond 16 tokens run as groups of 16 over gridDim.z
constexpr int DC_UMAX = 8;                       // units of a wave (chunks x fragments): 1 KiB of staging each
constexpr int DC_SSQ_LOADS = 16;                // per-fragment sums of squares a thread adds for its token: hidden <= 16 * 16 * threads / 16
constexpr int DC_STAGE_BYTES = DC_MAX_WAVES * DC_UMAX * 1024;
// A token's KiB of a chunk in LDS, padded by two 16-byte slots: ds_read_b128 serves lanes {0-3, 12-15, 20-27} (and
// the three like groups) in one cycle each, i.e. 8 tokens c at k-octet q and the other 8 at q + 1; with rows
// 2 slots (mod 16) apart those 16 lanes read slots 2 c + q, all different mod 16: no bank conflict.
constexpr int DC_XROW_BYTES = 1056;
// LDS of a block of `waves` waves with a codebook image of img_bytes: image + staging, or the activation regions
constexpr size_t dc_lds_main_bytes(int waves, size_t img_bytes) {
    const size_t a = img_bytes + (size_t)waves * DC_UMAX * 1024, b = (size_t)waves * 16 * DC_XROW_BYTES;
    return a > b ? a : b;
}
constexpr size_t dc_lds_bytes(int waves, size_t img_bytes) { return dc_lds_main_bytes(waves, img_bytes) + 512; }

struct DecodeParams {
    const unsigned char *x;        // [T][x_stride] act dtype
    const unsigned char *codes;    // packed pieces
    const unsigned char *codebooks;
    const uint16_t *scales;
    const uint16_t *bias;          // nullable
    uint16_t *y;
    int64_t y_stride;              // elements
    uint32_t x_stride_bytes;
    int T;                         // tokens of the batch (16 per block: gridDim.z groups)
    int nchunks;                   // 512-feature chunks of K (at most 1 << wk_log2)
    int K;                         // codebook entries (power of two, 16 .. 256)
    int nf;                        // fragments per wave
    int wk_log2;                   // log2 of the waves along K
    uint32_t img_bytes;            // K * 256, the staging area follows
    int spt;                       // 16-byte image slots per thread: max(1, K * 16 / threads per block)
    // K split over blocks (longer K than 8 chunks): block (x, y) takes chunks y * cpb .. + cpb - 1 of tile x; the fp32
    // partial sums [ksplit][T][n_total] meet in the workspace (the streaming kernel's protocol and layout)
    uint32_t aux_off;              // LDS offset of 512 bytes behind everything else (MODE 2: per-wave sums of squares)
    int ksplit;                    // blocks per tile = gridDim.y (1: no split, no workspace)
    int cpb;                       // chunks per block (at most 1 << wk_log2)
    int n_total;                   // output rows of the whole layer
    float *ws_part;
    unsigned *ws_cnt;              // one arrival counter per tile, zero between launches
    // MODE 1 (the GEMM in front of a fused add + RMSNorm): r = y + residual is written back as the new residual,
    // r * norm_w goes to xw, and the sum of squares of r per 16-row fragment and token to ssq[fragment][t_pad]
    uint16_t *residual;            // [T][res_stride], read and written
    int64_t res_stride;
    const uint16_t *norm_w;        // [n_total]
    uint16_t *xw;                  // [T][xw_stride]
    int64_t xw_stride;
    float *ssq;                    // MODE 1: written; MODE 2: read, [ssq_frags][t_pad]
    int t_pad;                     // tokens rounded up to 16
    // MODE 2 (the GEMM behind it): x holds r * norm_w; y is multiplied by rsqrt(sum of r^2 / hidden + eps) per token
    int ssq_frags;                 // fragments summed per token (hidden / 16)
    float eps, inv_hidden;
    // ACT (gate_up_proj with its activation): the layer is two equal partitions of act_rows rows (gate, up); tmap
    // covers the gate partition; a wave takes nf gate fragments and then the same rows' nf up fragments (2 nf units),
    // and y [T][act_rows] = silu(gate) * up.  `phases` is 2 (a run-time value, so that the phase loop stays a loop).
    int act_rows;
    int phases;
    TileMap tmap;                  // tiles of (blockDim/64 >> wk_log2) * nf * 16 rows
};

// ---- hand-issued memory instructions -------------------------------------------------------------------------
// The compiler tracks none of these; every wait is counted by hand.  "; PQX" marks a load or LDS read whose
// register stays in flight across branches and loop back edges, "; PQL" the point from which the register may be
// read: tools/lint_inflight.py checks in the generated assembly that no instruction in between, in text order,
// touches it.
template <int OFF>
__device__ __forceinline__ u32x4 dc_gload16(uint32_t voff, const unsigned char *sbase) {
    u32x4 v;
    asm volatile("; PQX %0\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ u32x2 dc_gload8(uint32_t voff, const void *sbase) {
    u32x2 v;
    asm volatile("; PQX %0\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
// 1 KiB from global memory straight into LDS at `lds_addr` (wave-uniform) + lane * 16
__device__ __forceinline__ void dc_dma16(uint32_t lds_addr, uint32_t voff, const unsigned char *sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ u32x4 dc_lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("; PQX %0\n\tds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 dc_lds_read16_at(uint32_t addr) {
    u32x4 v;
    asm volatile("; PQX %0\n\tds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <typename V>
__device__ __forceinline__ void dc_landed(V &v) { asm volatile("; PQL %0" : "+v"(v)); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate; n above 47 waits for 47):
// a computed jump into a table of 48 {s_waitcnt vmcnt(k); s_branch end} pairs of 8 bytes each.  s_getpc_b64 returns
// the address of the instruction behind it, 12 bytes in front of the table.
#define DC_VM_ENTRY(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 9f\n\t"
__device__ __forceinline__ void dc_wait_vm(int n) {
    asm volatile(
        "s_min_u32 s98, %0, 47\n\t"
        "s_lshl_b32 s98, s98, 3\n\t"
        "s_add_u32 s98, s98, 12\n\t"
        "s_getpc_b64 s[96:97]\n\t"
        "s_add_u32 s96, s96, s98\n\t"
        "s_addc_u32 s97, s97, 0\n\t"
        "s_setpc_b64 s[96:97]\n\t"
        DC_VM_ENTRY(0) DC_VM_ENTRY(1) DC_VM_ENTRY(2) DC_VM_ENTRY(3) DC_VM_ENTRY(4) DC_VM_ENTRY(5) DC_VM_ENTRY(6) DC_VM_ENTRY(7)
        DC_VM_ENTRY(8) DC_VM_ENTRY(9) DC_VM_ENTRY(10) DC_VM_ENTRY(11) DC_VM_ENTRY(12) DC_VM_ENTRY(13) DC_VM_ENTRY(14) DC_VM_ENTRY(15)
        DC_VM_ENTRY(16) DC_VM_ENTRY(17) DC_VM_ENTRY(18) DC_VM_ENTRY(19) DC_VM_ENTRY(20) DC_VM_ENTRY(21) DC_VM_ENTRY(22) DC_VM_ENTRY(23)
        DC_VM_ENTRY(24) DC_VM_ENTRY(25) DC_VM_ENTRY(26) DC_VM_ENTRY(27) DC_VM_ENTRY(28) DC_VM_ENTRY(29) DC_VM_ENTRY(30) DC_VM_ENTRY(31)
        DC_VM_ENTRY(32) DC_VM_ENTRY(33) DC_VM_ENTRY(34) DC_VM_ENTRY(35) DC_VM_ENTRY(36) DC_VM_ENTRY(37) DC_VM_ENTRY(38) DC_VM_ENTRY(39)
        DC_VM_ENTRY(40) DC_VM_ENTRY(41) DC_VM_ENTRY(42) DC_VM_ENTRY(43) DC_VM_ENTRY(44) DC_VM_ENTRY(45) DC_VM_ENTRY(46) DC_VM_ENTRY(47)
        "9:"
        ::"s"(n) : "memory", "s96", "s97", "s98", "scc");
}
#undef DC_VM_ENTRY


// STAMPS (diagnostic builds only): thread 0 writes wall-clock stamps of the phases to `bias` (8 x u64 per block)
// ABL (diagnostic builds only; 0 in the library): bit 0 leaves the MFMAs of the units out, bit 1 the gathers
// MODE: 0 the plain linear; 1 and 2 the two halves of a fused add + RMSNorm between two linears (DecodeParams)
// ACT: gate_up_proj with SiLU-and-mul in the epilogue (DecodeParams; MODE 0 or 2; K = 256 entries, 8 waves, no K split):
// the block computes its gate fragments with the gate codebook's image, swaps the image for the up codebook's and
// computes the same rows' up fragments -- one codebook image at a time is all the LDS holds
template <typename AT, bool STAMPS = false, int ABL = 0, int MODE = 0, bool ACT = false>
__global__ __launch_bounds__(DC_MAX_WAVES * 64) void gemm_decode_kernel(const DecodeParams p) {
    typedef typename Act<AT>::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint16_t *bias = p.bias;
    // batches of more than 16 tokens: block z takes tokens 16 z .. 16 z + 15 (every group gathers the weights again)
    const int tok0 = (int)blockIdx.z * 16;
    const int T = p.T - tok0 < 16 ? p.T - tok0 : 16;          // tokens of this block
    // STAMPS: the first and the last wave keep 12 wall-clock stamps each in registers and store them at the very end
    // (24 x u64 per block behind `bias`): a store in between would sit in the memory pipeline with the loads
    unsigned long long st[12];
    if constexpr (STAMPS) {
        bias = nullptr;
#pragma unroll
        for (int k = 0; k < 12; ++k) st[k] = 0;
        st[0] = __builtin_amdgcn_s_memrealtime();
    }
#define DC_STAMP(k) if constexpr (STAMPS) { asm volatile("" ::: "memory"); st[k] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); }

    // A launch of a few microseconds is paced by the instructions its waves issue (a wave issues one every 4-5
    // cycles at best, the two waves of a SIMD share the scalar unit, and code that runs once is fetched at a
    // third of that rate) and by two data streams: the code pieces from HBM, and the activations from L2 --
    // every block reads all of x, 32 MB of L2 reads for 256 blocks at 16 tokens and K = 4096.
    //
    // How the activations come in decides the first microseconds.  MFMA k-step s wants lane (q = l >> 4, c = l & 15)
    // to hold x[token c][32 s + 8 q .. + 7]; a load shaped like that has the four lanes of every quad on four
    // different rows, and the memory pipeline takes such a wave-load at a quarter of the rate of one whose quads each
    // read 64 contiguous bytes (measured, tools/ingest_lab.hip: 128 KiB per block in 4.0 us against 1.6).  So a wave
    // brings its chunk of every token in as ONE contiguous KiB per token, by LDS-DMA, into a padded LDS region of its
    // own (16 rows of 1056 bytes: the padding puts the lanes one LDS cycle serves on different banks), and reads
    // the 16 operand registers from there (lane (q, c): row c, byte 64 s + 16 q).  The region lies where the codebook
    // image and the code staging area are built AFTERWARDS: the activations are in registers by then.
    const int WK = 1 << p.wk_log2;
    const int wk = wave & (WK - 1), wr = wave >> p.wk_log2;
    const int nf = p.nf;
    const int chunk0 = (int)blockIdx.y * p.cpb;               // first chunk of this block's K range
    const int nck_blk = p.nchunks - chunk0 < p.cpb ? p.nchunks - chunk0 : p.cpb;   // chunks in it (at least one)
    const int nck = wk < nck_blk ? 1 : 0;                     // a wave beyond the last chunk has no work
    const int NU = ACT ? 2 * nf : nf;                         // units of a wave with a chunk (at most DC_UMAX)
    const int U = nck * NU;

    // ---- the block's tile
    const int frags_blk = (int)((blockDim.x >> 6) >> p.wk_log2) * nf;
    const TileLoc loc = locate_tile(p.tmap, (int)blockIdx.x, frags_blk * 16);
    const int nfr = loc.nrows >> 4;                       // valid fragments of the tile (partitions are multiples of 16 rows)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t stage0 = lds0 + p.img_bytes + (uint32_t)wave * (DC_UMAX * 1024);   // this wave's staging slots
    const uint32_t lane16 = (uint32_t)lane * 16u;
    DC_STAMP(1)   // tile found

    // ---- codebook entries: entry e fills 16 image slots of 16 bytes (one per lane residue); a thread owns spt
    // consecutive slots.  Requested first, stored once the activations are out of the LDS.
    int issued = 0;
    const int spt = p.spt;
    const bool img_active = tid * spt < (p.K << 4);
    const int ncb = spt > 16 ? spt >> 4 : 1;          // entries per thread (blocks of one or two waves: up to 4)
    const int e0 = img_active ? (tid * spt) >> 4 : 0;
    const unsigned char *cb_part = p.codebooks + (size_t)loc.part * p.K * 16;
    u32x4 cbv0 = dc_gload16<0>((uint32_t)e0 * 16u, cb_part), cbv1, cbv2, cbv3;
    if (ncb > 1) {   // always four loads then (a block of two waves re-reads its two entries: same addresses, no traffic)
        cbv1 = dc_gload16<16>((uint32_t)e0 * 16u, cb_part);
        cbv2 = dc_gload16<0>((uint32_t)(e0 + (ncb > 2 ? 2 : 0)) * 16u, cb_part);
        cbv3 = dc_gload16<16>((uint32_t)(e0 + (ncb > 2 ? 2 : 0)) * 16u, cb_part);
        issued += 4;
    } else {
        ++issued;
    }
    u32x4 cbu0{0u, 0u, 0u, 0u};   // ACT: the same entry of the up partition's codebook (K = 256, 512 threads: one entry a thread)
    if constexpr (ACT) {
        cbu0 = dc_gload16<0>((uint32_t)e0 * 16u, cb_part + (size_t)p.K * 16);
        ++issued;
    }

    // MODE 2: this thread's share of the per-fragment sums of squares of its token (tid & 15): fragments tid >> 4,
    // + threads / 16, ... (at most DC_SSQ_LOADS: checked on the host), requested here and added once the activations
    // have landed
    float ssq_v[DC_SSQ_LOADS];
    if constexpr (MODE == 2) {
        const int fstep = (int)(blockDim.x >> 4);
        const float *sp = p.ssq + tok0 + (tid & 15);
#pragma unroll
        for (int j = 0; j < DC_SSQ_LOADS; ++j) {
            const int f = (tid >> 4) + j * fstep;
            ssq_v[j] = sp[(size_t)(f < p.ssq_frags ? f : p.ssq_frags - 1) * p.t_pad];
        }
    }

    // ---- activations: token t's KiB of the wave's chunk -> row t of the wave's region.  With at most 8 tokens only
    // 8 rows are brought in (columns 8..15 of the MFMAs then read whatever the LDS holds: those outputs are never
    // stored, and a column's values do not reach another column).  Tokens past T repeat the last one.
    const uint32_t xreg = lds0 + (uint32_t)wave * (16u * DC_XROW_BYTES);
    const unsigned char *xb0 = p.x + (size_t)tok0 * p.x_stride_bytes + (size_t)(chunk0 + (wk < nck_blk ? wk : nck_blk - 1)) * 1024;
    const int ntok = T > 8 ? 16 : 8;
    {
        uint32_t xdst = xreg;
#pragma nounroll
        for (int t = 0; t < ntok; ++t) {
            const uint32_t tc = (uint32_t)(t < T ? t : T - 1);
            dc_dma16(xdst, lane16, xb0 + (size_t)(tc * p.x_stride_bytes));
            xdst += DC_XROW_BYTES;
        }
    }
    issued += ntok;
    DC_STAMP(2)   // entries and activations requested

    // Unit i is chunk wk of fragment wr * nf + i of the tile (fragments past the tile's last repeat it and are
    // never stored); its piece goes by DMA straight into staging slot i of the wave.
    int fb = wr * nf, nvalid = nfr - fb;           // first fragment of the wave, fragments it really has
    if (nvalid < 1) { fb = nfr - 1; nvalid = 1; }
    const size_t frag_bytes = (size_t)p.nchunks * 1024;
    const unsigned char *dma_base = p.codes + ((size_t)((loc.row0 >> 4) + fb) * p.nchunks + chunk0 + wk) * 1024;
    const size_t up_bytes = ACT ? (size_t)(p.act_rows >> 4) * frag_bytes : 0;   // from a gate fragment to the same rows' up fragment
    auto issue_dma = [&](int j) {   // unit j: fragment j (ACT: units nf .. 2 nf - 1 are the up fragments of units 0 .. nf - 1)
        if (j < U) {
            const int jj = ACT && j >= nf ? j - nf : j;
            const size_t off = (size_t)(jj < nvalid ? jj : nvalid - 1) * frag_bytes + (ACT && j >= nf ? up_bytes : 0);
            dc_dma16(stage0 + (uint32_t)j * 1024u, lane16, dma_base + off);
        }
    };

    // ---- the activations have landed (and so have the codebook entries, which are older): operand registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dc_landed(cbv0);
    if constexpr (ACT) dc_landed(cbu0);
    issued = 0;
    DC_STAMP(3)   // entries and activations landed
    u32x4 B0[16];
    {
        const uint32_t xrd = xreg + (uint32_t)(lane & 15) * DC_XROW_BYTES + (uint32_t)(lane >> 4) * 16u;
#define DC_XR(s) B0[s] = dc_lds_read16_at<64 * (s)>(xrd);
        DC_XR(0) DC_XR(1) DC_XR(2) DC_XR(3) DC_XR(4) DC_XR(5) DC_XR(6) DC_XR(7)
        DC_XR(8) DC_XR(9) DC_XR(10) DC_XR(11) DC_XR(12) DC_XR(13) DC_XR(14) DC_XR(15)
#undef DC_XR
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define DC_XD(s) dc_landed(B0[s]);
        DC_XD(0) DC_XD(1) DC_XD(2) DC_XD(3) DC_XD(4) DC_XD(5) DC_XD(6) DC_XD(7)
        DC_XD(8) DC_XD(9) DC_XD(10) DC_XD(11) DC_XD(12) DC_XD(13) DC_XD(14) DC_XD(15)
#undef DC_XD
    }
    DC_STAMP(4)   // operands in registers
    if constexpr (MODE == 2) {
        // The token's sum over all fragments in an order that does not depend on the block's wave count (so a token's
        // 1 / rms is the same number whatever geometry its batch size picked): the order of a block of 8 waves.  Row
        // tid >> 4 of such a block adds fragments row, row + 32, ...; a wave adds its four rows; the epilogue adds the
        // 8 waves' sums in order.  A block of 8 / R waves plays R of those waves per wave: its k-th load belongs to
        // row (tid >> 4) + (k mod R) * rows of the block.
        const int fstep = (int)(blockDim.x >> 4);
        float v[DC_SSQ_LOADS];
#pragma unroll
        for (int j = 0; j < DC_SSQ_LOADS; ++j) {
            asm volatile("" : "+v"(ssq_v[j]));      // used from here on: behind the wait for the activations
            v[j] = ((tid >> 4) + j * fstep < p.ssq_frags) ? ssq_v[j] : 0.f;
        }
        float *aux = reinterpret_cast<float *>(smem + p.aux_off);
        auto reduce = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
#pragma unroll
                for (int k = r; k < DC_SSQ_LOADS; k += R) s += v[k];
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16) aux[(wave + r * (8 / R)) * 16 + lane] = s;
            }
        };
        switch (blockDim.x >> 6) {
        case 8: reduce(std::integral_constant<int, 1>{}); break;
        case 4: reduce(std::integral_constant<int, 2>{}); break;
        case 2: reduce(std::integral_constant<int, 4>{}); break;
        default: reduce(std::integral_constant<int, 8>{}); break;
        }
    }
    __syncthreads();   // every wave has its activations out of the LDS: image and staging may be written
    DC_STAMP(5)   // barrier

    // ---- code pieces (HBM: the longest latency, requested first), epilogue operands, codebook image
    const int seq_dma0 = issued;
#pragma nounroll
    for (int j = 0; j < U; ++j) issue_dma(j);
    issued += U;
    // wave (wk, wr) with wk < nf finishes fragment wk of row group wr: token lane & 15, rows 4 (lane >> 4) .. + 3
    const int out_frag = wr * nf + wk;
    const bool out_wave = wk < nf && out_frag < nfr;
    const int out_row = loc.row0 + (out_wave ? out_frag : 0) * 16 + (lane >> 4) * 4;
    u32x2 scv = dc_gload8((uint32_t)out_row * 2u, p.scales);
    ++issued;
    u32x2 biv = dc_gload8((uint32_t)out_row * 2u, bias ? bias : p.scales);   // without a bias: loaded, never used
    ++issued;
    u32x2 scu{0u, 0u}, biu{0u, 0u};   // ACT: scales and bias of the up rows
    if constexpr (ACT) {
        scu = dc_gload8((uint32_t)(out_row + p.act_rows) * 2u, p.scales);
        ++issued;
        biu = dc_gload8((uint32_t)(out_row + p.act_rows) * 2u, bias ? bias : p.scales);
        ++issued;
    }
    u32x2 rsv{0u, 0u}, nwv{0u, 0u};
    if constexpr (MODE == 1) {   // the lane's four residual values (its token, its rows) and norm weights
        const int tl = (lane & 15) < T ? (lane & 15) : T - 1;
        rsv = dc_gload8((uint32_t)((int64_t)(tok0 + tl) * p.res_stride + out_row) * 2u, p.residual);
        ++issued;
        nwv = dc_gload8((uint32_t)out_row * 2u, p.norm_w);
        ++issued;
    }
    DC_STAMP(6)   // everything requested
    if (img_active) {
        if (spt == 8) {
            // 256 entries, 512 threads: the thread's 8 slots, in an order permuted per lane (slot c ^ (tid & 7)) so
            // that the lanes of a store group land on different banks
            const uint32_t a0 = (uint32_t)tid * 128u + ((uint32_t)(tid & 7) << 4);
#define DC_IS(c) *reinterpret_cast<u32x4 *>(smem + (a0 ^ ((c) << 4))) = cbv0;
            DC_IS(0) DC_IS(1) DC_IS(2) DC_IS(3) DC_IS(4) DC_IS(5) DC_IS(6) DC_IS(7)
#undef DC_IS
        } else {
            const int spe = spt > 16 ? 16 : spt;              // slots of one entry that this thread writes
            auto put = [&](const u32x4 &v, int base) {
#pragma nounroll
                for (int c = 0; c < spe; ++c)
                    *reinterpret_cast<u32x4 *>(smem + (size_t)(base + ((c + tid) & (spe - 1))) * 16) = v;
            };
            const int first = tid * spt;
            put(cbv0, first);
            if (ncb > 1) {
                dc_landed(cbv1); dc_landed(cbv2); dc_landed(cbv3);
                put(cbv1, first + 16);
                if (ncb > 2) {
                    put(cbv2, first + 32);
                    put(cbv3, first + 48);
                }
            }
        }
    }
    if (nck == 0) {   // a wave without a chunk (more waves along K than chunks) contributes zeros
#pragma nounroll
        for (int i = 0; i < NU; ++i)
            *reinterpret_cast<f32x4 *>(smem + p.img_bytes + ((size_t)(wave * DC_UMAX + i) * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // the image is complete
    DC_STAMP(7)   // image stored, barrier

    // ---- units ------------------------------------------------------------------------------------------------------
    const uint32_t img_lane = lds0 + (uint32_t)(lane & 15) * 16u;   // this lane's copy of entry 0
    const uint32_t sel0 = 0x0c0c0400u, sel1 = 0x0c0c0500u, sel2 = 0x0c0c0600u, sel3 = 0x0c0c0700u;
    // The wave's units, software-pipelined over two sets of 8 gathered weight fragments: while the MFMAs of one
    // half of a unit's k-steps run, the gathers of the other half (or of the next unit's first half) are in flight,
    // so the LDS and the matrix pipe work at the same time.  LDS reads return in order; every count below is the
    // number of reads issued behind the one waited for.
    auto units = [&](const u32x4 (&B)[16], int i0, int i1) {
        u32x4 a0[8], a1[8];
        dc_wait_vm(issued - 1 - (seq_dma0 + i0));
        u32x4 cw = dc_lds_read16(stage0 + (uint32_t)i0 * 1024u + lane16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dc_landed(cw);
#define DC_G(A, j, s, sel) if constexpr (ABL & 2) { A[j] = cw; asm volatile("; PQX %0" : "+v"(A[j])); } else A[j] = dc_lds_read16(__builtin_amdgcn_perm(cw[(s) >> 2], img_lane, sel));
#define DC_GLO DC_G(a0, 0, 0, sel0) DC_G(a0, 1, 1, sel1) DC_G(a0, 2, 2, sel2) DC_G(a0, 3, 3, sel3) DC_G(a0, 4, 4, sel0) DC_G(a0, 5, 5, sel1) DC_G(a0, 6, 6, sel2) DC_G(a0, 7, 7, sel3)
#define DC_GHI DC_G(a1, 0, 8, sel0) DC_G(a1, 1, 9, sel1) DC_G(a1, 2, 10, sel2) DC_G(a1, 3, 11, sel3) DC_G(a1, 4, 12, sel0) DC_G(a1, 5, 13, sel1) DC_G(a1, 6, 14, sel2) DC_G(a1, 7, 15, sel3)
#define DC_MW(j, s, n)                                                                    \
    asm volatile("s_waitcnt lgkmcnt(%1)\n\t; PQL %0" : "+v"(a0[j]) : "n"(n));               \
    if constexpr (ABL & 1) acc.x += __uint_as_float(a0[j].x ^ B[s].x); else                \
    acc = Act<AT>::mfma16(__builtin_bit_cast(vec8, a0[j]), __builtin_bit_cast(vec8, B[s]), acc); \
    __builtin_amdgcn_sched_barrier(0);
#define DC_MN(j, s)                                                                       \
    if constexpr (ABL & 1) acc.x += __uint_as_float(a1[j].x ^ B[s].x); else                \
    acc = Act<AT>::mfma16(__builtin_bit_cast(vec8, a1[j]), __builtin_bit_cast(vec8, B[s]), acc); \
    __builtin_amdgcn_sched_barrier(0);
        DC_GLO
#pragma nounroll
        for (int i = i0; i < i1; ++i) {
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
            DC_GHI
            // the next unit's codes (the last unit reads its own again: same counts, no branch)
            const int in = i + 1 < i1 ? i + 1 : i;
            dc_wait_vm(issued - 1 - (seq_dma0 + in));
            u32x4 cwn = dc_lds_read16(stage0 + (uint32_t)in * 1024u + lane16);
            // behind a0[j]: 7 - j of its set, the store of the unit before, 8 of the other set, the read above
            DC_MW(0, 0, 15) DC_MW(1, 1, 15) DC_MW(2, 2, 14) DC_MW(3, 3, 13) DC_MW(4, 4, 12) DC_MW(5, 5, 11) DC_MW(6, 6, 10) DC_MW(7, 7, 9)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dc_landed(cwn);
#define DC_L1(j) dc_landed(a1[j]);
            DC_L1(0) DC_L1(1) DC_L1(2) DC_L1(3) DC_L1(4) DC_L1(5) DC_L1(6) DC_L1(7)
#undef DC_L1
            cw = cwn;
            DC_GLO                       // first half of the next unit (of this one again after the last: never used)
            DC_MN(0, 8) DC_MN(1, 9) DC_MN(2, 10) DC_MN(3, 11) DC_MN(4, 12) DC_MN(5, 13) DC_MN(6, 14) DC_MN(7, 15)
            *reinterpret_cast<f32x4 *>(smem + p.img_bytes + ((size_t)(wave * DC_UMAX + i) * 64 + lane) * 16) = acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the gathers issued behind the last unit
#define DC_L0(j) dc_landed(a0[j]);
        DC_L0(0) DC_L0(1) DC_L0(2) DC_L0(3) DC_L0(4) DC_L0(5) DC_L0(6) DC_L0(7)
#undef DC_L0
#undef DC_G
#undef DC_GLO
#undef DC_GHI
#undef DC_MW
#undef DC_MN
    };
    if constexpr (ACT) {
#pragma nounroll
        for (int ph = 0; ph < p.phases; ++ph) {
            if (ph) {   // every wave is past its last gather of the gate image: the up codebook takes its place
                __syncthreads();
                if (img_active) {
                    const uint32_t a0 = (uint32_t)tid * 128u + ((uint32_t)(tid & 7) << 4);
#define DC_IS(c) *reinterpret_cast<u32x4 *>(smem + (a0 ^ ((c) << 4))) = cbu0;
                    DC_IS(0) DC_IS(1) DC_IS(2) DC_IS(3) DC_IS(4) DC_IS(5) DC_IS(6) DC_IS(7)
#undef DC_IS
                }
                __syncthreads();
            }
            if (nck > 0) units(B0, ph * nf, ph * nf + nf);
        }
    } else {
        if (nck > 0) units(B0, 0, nf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dc_landed(scv);
    dc_landed(biv);
    if constexpr (MODE == 1) { dc_landed(rsv); dc_landed(nwv); }
    if constexpr (ACT) { dc_landed(scu); dc_landed(biu); }
    DC_STAMP(9)   // this wave's units done
    __syncthreads();
    DC_STAMP(10)   // all waves' units done

    // ---- epilogue: the partial sums of the output fragment over the waves along K, in a fixed order -------------------
    const int t_out = lane & 15;
    float rinv = 1.f;
    if constexpr (MODE == 2) {   // 1 / rms of the token: the 8 partial sums in order
        const float *wp = reinterpret_cast<const float *>(smem + p.aux_off) + t_out;
        float ss = wp[0];
#pragma nounroll
        for (int w = 1; w < 8; ++w) ss += wp[w * 16];
        rinv = rsqrtf(ss * p.inv_hidden + p.eps);
    }
    float ss_lane = 0.f;         // MODE 1: sum of squares of this lane's four values of r
    auto finish = [&](const f32x4 &sum) {   // scale, bias, one rounding, 8-byte store (y and its row stride are 8-byte aligned: host)
        const uint32_t sw[2] = {scv.x, scv.y}, bw[2] = {biv.x, biv.y}, rw[2] = {rsv.x, rsv.y}, nw2[2] = {nwv.x, nwv.y};
        uint16_t h[4], hx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t s16 = (uint16_t)(sw[e >> 1] >> ((e & 1) * 16)), b16 = (uint16_t)(bw[e >> 1] >> ((e & 1) * 16));
            float v = (MODE == 2 ? sum[e] * rinv : sum[e]) * Act<AT>::to_f32(s16) + (bias ? Act<AT>::to_f32(b16) : 0.0f);
            if constexpr (MODE == 1) {
                v += Act<AT>::to_f32((uint16_t)(rw[e >> 1] >> ((e & 1) * 16)));
                ss_lane += v * v;
                hx[e] = Act<AT>::from_f32(v * Act<AT>::to_f32((uint16_t)(nw2[e >> 1] >> ((e & 1) * 16))));
            }
            h[e] = Act<AT>::from_f32(v);
        }
        u32x2 o2;
        o2.x = h[0] | ((uint32_t)h[1] << 16);
        o2.y = h[2] | ((uint32_t)h[3] << 16);
        if constexpr (MODE == 1) {
            *reinterpret_cast<u32x2 *>(p.residual + (size_t)(tok0 + t_out) * p.res_stride + out_row) = o2;
            u32x2 o3;
            o3.x = hx[0] | ((uint32_t)hx[1] << 16);
            o3.y = hx[2] | ((uint32_t)hx[3] << 16);
            *reinterpret_cast<u32x2 *>(p.xw + (size_t)(tok0 + t_out) * p.xw_stride + out_row) = o3;
        } else {
            *reinterpret_cast<u32x2 *>(p.y + (size_t)(tok0 + t_out) * p.y_stride + out_row) = o2;
        }
    };
    // MODE 1, by every wave that finished a fragment (wave-uniform condition): the fragment's sum of squares per token
    auto put_ssq = [&]() {
        if constexpr (MODE == 1) {
            float ss = ss_lane;
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (lane < T) p.ssq[(size_t)(out_row >> 4) * p.t_pad + tok0 + lane] = ss;
        }
    };
    if constexpr (ACT) {
        // wave wk < nf finishes gate fragment wk and up fragment wk of its row group: y = silu(g) * u, every factor
        // rounded to the act dtype first, exactly as the GEMM followed by silu_and_mul rounds them
        if (out_wave) {
            const unsigned char *slot = smem + p.img_bytes + ((size_t)((wr << p.wk_log2) * DC_UMAX + wk) * 64 + lane) * 16;
            f32x4 sg = *reinterpret_cast<const f32x4 *>(slot), su = *reinterpret_cast<const f32x4 *>(slot + (size_t)nf * 1024);
#pragma nounroll
            for (int w = 1; w < WK; ++w) {
                const f32x4 vg = *reinterpret_cast<const f32x4 *>(slot + (size_t)w * (DC_UMAX * 1024));
                const f32x4 vu = *reinterpret_cast<const f32x4 *>(slot + (size_t)w * (DC_UMAX * 1024) + (size_t)nf * 1024);
                sg.x += vg.x; sg.y += vg.y; sg.z += vg.z; sg.w += vg.w;
                su.x += vu.x; su.y += vu.y; su.z += vu.z; su.w += vu.w;
            }
            if (t_out < T) {
                const uint32_t sw[2] = {scv.x, scv.y}, bw[2] = {biv.x, biv.y}, swu[2] = {scu.x, scu.y}, bwu[2] = {biu.x, biu.y};
                uint16_t h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sh = (e & 1) * 16;
                    const float g = (MODE == 2 ? sg[e] * rinv : sg[e]) * Act<AT>::to_f32((uint16_t)(sw[e >> 1] >> sh)) +
                                    (bias ? Act<AT>::to_f32((uint16_t)(bw[e >> 1] >> sh)) : 0.0f);
                    const float u = (MODE == 2 ? su[e] * rinv : su[e]) * Act<AT>::to_f32((uint16_t)(swu[e >> 1] >> sh)) +
                                    (bias ? Act<AT>::to_f32((uint16_t)(bwu[e >> 1] >> sh)) : 0.0f);
                    const float yg = Act<AT>::to_f32(Act<AT>::from_f32(g)), yu = Act<AT>::to_f32(Act<AT>::from_f32(u));
                    const float sl = Act<AT>::to_f32(Act<AT>::from_f32(yg / (1.0f + expf(-yg))));
                    h[e] = Act<AT>::from_f32(sl * yu);
                }
                u32x2 o2;
                o2.x = h[0] | ((uint32_t)h[1] << 16);
                o2.y = h[2] | ((uint32_t)h[3] << 16);
                *reinterpret_cast<u32x2 *>(p.y + (size_t)(tok0 + t_out) * p.y_stride + out_row) = o2;
            }
        }
    } else if (out_wave) {
        const unsigned char *slot = smem + p.img_bytes + ((size_t)((wr << p.wk_log2) * DC_UMAX + wk) * 64 + lane) * 16;
        f32x4 sum = *reinterpret_cast<const f32x4 *>(slot);
#pragma nounroll
        for (int w = 1; w < WK; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(slot + (size_t)w * (DC_UMAX * 1024));
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        if (t_out < T) {
            if (p.ksplit == 1) finish(sum);
            else coherent_store16(p.ws_part + ((size_t)blockIdx.y * p.T + tok0 + t_out) * (size_t)p.n_total + out_row, sum);
        }
        if (p.ksplit == 1) put_ssq();
    }
    // K split over blocks: every block counts itself in on its tile's counter; the last one to arrive adds the partial
    // sums up in split order (the result does not depend on which block that is), finishes the tile and leaves the
    // counter at zero.  Partials and counter go past the per-XCD L2 (device-scope relaxed accesses), ordered by waiting
    // for the stores' completion before the count.
    if (p.ksplit > 1) {
        // the flag lives in the first word of the staging area (every accumulator slot has been read before the
        // barrier below); a static __shared__ variable would move the image off its 64 KiB-aligned base
        volatile int &s_last = *reinterpret_cast<volatile int *>(smem + p.img_bytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned *cnt = p.ws_cnt + blockIdx.z * gridDim.x + blockIdx.x;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)p.ksplit - 1u;
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last && out_wave) {
          if (t_out < T) {
            f32x4 part[8];   // all requested before any is waited for; splits past the last re-read the last one
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const int zc = z < p.ksplit ? z : p.ksplit - 1;
                part[z] = coherent_load16_issue(p.ws_part + ((size_t)zc * p.T + tok0 + t_out) * (size_t)p.n_total + out_row);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int z = 0; z < 8; ++z) coherent_landed(part[z]);
            f32x4 sum{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                if (z < p.ksplit) { sum.x += part[z].x; sum.y += part[z].y; sum.z += part[z].z; sum.w += part[z].w; }
            }
            finish(sum);
          }
          put_ssq();
        }
    }
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[11] = __builtin_amdgcn_s_memrealtime();   // stores retired
        const int nw = (int)(blockDim.x >> 6);
        if (lane == 0 && (wave == 0 || wave == nw - 1)) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<uint16_t *>(p.bias)) +
                                    (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 24 + (wave == 0 ? 0 : 12);
#pragma unroll
            for (int k = 0; k < 12; ++k) o[k] = st[k];
        }
    }
#undef DC_STAMP
}

}  // namespace pq
