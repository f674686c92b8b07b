// Fused codebook gather + MFMA GEMM for large token counts (prefill) on gfx950.
//
// y[t][n] = scale[n] * sum_k x[t][k] * W[n][k] + bias[n]
//
// A block of 8 waves owns a BN-row (n) x BT-token output tile and walks K in steps of 64.  The
// weight is never staged: it is the MFMA A operand, and lane l's fragment (row l&15, k = 8(l>>4)..+7)
// is one d=8 centroid (two d=4, four d=2), read with one ds_read_b128 from the replicated LDS
// codebook image at the address the lane's code selects (see pq_gemm_stream_impl.cuh).  Only the
// activations go through LDS: [BT][64] tiles, written by LDS-DMA into a ring of 3 stages of KT tiles each
// (32 KiB a stage whatever the tile shape: small token tiles take more k per stage, so that as many
// bytes are in flight and a barrier is as rare as with the large ones), XOR-swizzled so that the
// B-operand reads (16 tokens x 4 k-octets per wave-instruction) are bank-conflict free.  A gathered
// A fragment is reused for WT token fragments, an activation fragment for WN weight fragments.
//
// Codes come in the PACKED order (pq_gemm_stream.hip): per 16-row fragment and k-chunk one 16-byte
// piece per lane holding the codes of that lane's KS k-steps, loaded once per chunk.
//
// Tile -> block map: blocks b and b+8 share an XCD (and its L2).  Each XCD takes a contiguous run of
// (token tile, row tile) pairs, ordered in groups of 4 token tiles x consecutive row tiles, so the blocks
// resident on one XCD share a few activation tiles and a few row tiles' codes in that XCD's L2.
#pragma once
#include <stdlib.h>

#include <atomic>

#include <type_traits>
#include <utility>

#include "pq_common.cuh"
#include "pq_gather.cuh"
#include "pq_kernels.h"
#include "pq_knobs.h"
#include "pq_stream_geom.cuh"

namespace pq {

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ds_read_b128 by inline asm: the compiler does not track it, so the counted s_waitcnt lgkmcnt(n) placed by
// hand below are what is executed (its own waits for LDS reads are always lgkmcnt(0)).  OFF is the
// instruction's immediate byte offset.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("; PQI %0\n\tds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// The same for the 8- and 4-byte entries of the d = 4 and d = 2 formats.
__device__ __forceinline__ u32x2 lds_read8(uint32_t addr) {
    u32x2 v;
    asm volatile("; PQI %0\n\tds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
    uint32_t v;
    asm volatile("; PQI %0\n\tds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// A gathered weight fragment in flight: the lane's 16 operand bytes are one d = 8 entry, two d = 4 entries or four
// d = 2 entries, each read by an instruction of its own into its part of the fragment.
template <int D>
struct GatherFrag;
template <>
struct GatherFrag<8> {
    u32x4 v;
    __device__ __forceinline__ void read(const uint32_t (&a)[1]) { v = lds_read16<0>(a[0]); }
    __device__ __forceinline__ void landed() { asm volatile("; PQL %0" : "+v"(v)); }
    __device__ __forceinline__ u32x4 value() const { return v; }
};
template <>
struct GatherFrag<4> {
    u32x2 h[2];
    __device__ __forceinline__ void read(const uint32_t (&a)[2]) { h[0] = lds_read8(a[0]); h[1] = lds_read8(a[1]); }
    __device__ __forceinline__ void landed() { asm volatile("; PQL %0\n\t; PQL %1" : "+v"(h[0]), "+v"(h[1])); }
    __device__ __forceinline__ u32x4 value() const { return u32x4{h[0].x, h[0].y, h[1].x, h[1].y}; }
};
template <>
struct GatherFrag<2> {
    uint32_t w[4];
    __device__ __forceinline__ void read(const uint32_t (&a)[4]) {
        w[0] = lds_read4(a[0]); w[1] = lds_read4(a[1]); w[2] = lds_read4(a[2]); w[3] = lds_read4(a[3]);
    }
    __device__ __forceinline__ void landed() {
        asm volatile("; PQL %0\n\t; PQL %1\n\t; PQL %2\n\t; PQL %3" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    }
    __device__ __forceinline__ u32x4 value() const { return u32x4{w[0], w[1], w[2], w[3]}; }
};

// LDS address of this lane's copy of the d = 8 codebook entry that code `i` of a code piece selects.  One-byte
// codes (K <= 256) always get 16 copies, 256 bytes per entry; they are stored reduced mod K in the packed layout
// and the image starts at a multiple of 64 KiB, so the address is the code byte dropped into bits 8..15 of the
// lane's offset: one v_perm_b32 instead of a bit-field extract and a shift-add.
template <int CB, typename VecT>
__device__ __forceinline__ uint32_t gather_addr8(const VecT &w, int i, uint32_t nbits, uint32_t shift, uint32_t img_lane) {
    if constexpr (CB == 1) {
        return __builtin_amdgcn_perm(w[i >> 2], img_lane, 0x0c0c0000u | ((4u + (uint32_t)(i & 3)) << 8));
    } else {
        return img_lane + (code_at<CB>(w, i, nbits) << shift);
    }
}

// The hand-pipelined stage issues its LDS reads in this order: activations of k-step 0, gathers of groups 0 and 1
// (PRE: the stage before it has issued them already, gathers first); then group h (before and between its MFMAs): gathers of group h+2, activations of the next k-step when h is the
// first group of a k-step.  LDS reads return in order; this is the number of reads issued after the last one
// group g's MFMAs need (its gathers and, for the first group of a k-step, that k-step's activations) and before
// group g's wait: the n of its `s_waitcnt lgkmcnt(n)`.
template <int NG, int NPAIR, int NKS, int WT, bool PRE = false, int GR = 2>
constexpr int pipe_newer(int g) {   // GR: LDS reads per group's gathers (two fragments of 1, 2 or 4 entries)
    const int kk = g / NPAIR;
    const bool need_x = g % NPAIR == 0;
    int after = 0;          // reads issued after the last needed one so far
    if (PRE) {
        // issued by the stage before this one, in the gaps of its last group: G(0), G(1), X(0)
        if (g == 0) after = 0; else after += GR;
        if (NG > 1) { if (g == 1) after = 0; else after += GR; }
        if (need_x && kk == 0) after = 0; else after += WT;
    } else {
        // prologue: X(0), G(0), G(1)
        if (g == 0) after = 0; else after += GR;        // G(0) is newer than X(0)
        if (NG > 1) { if (g == 1) after = 0; else after += GR; }
    }
    for (int h = 0; h < g; ++h) {
        if (h + 2 < NG) { if (h + 2 == g) after = 0; else after += GR; }
        if (h % NPAIR == 0 && h / NPAIR + 1 < NKS) { if (need_x && h / NPAIR + 1 == kk) after = 0; else after += WT; }
    }
    return after;
}

// One dword of a code piece, requested by inline asm (W: which dword of the lane's 16 bytes).  ACROSS: the request
// stays in flight across branches ("; PQR" for the lint: nothing that can run between it and its landing may touch the register,
// whatever the control flow); otherwise no branch may come before it has landed ("; PQI").
template <int W, bool ACROSS>
__device__ __forceinline__ uint32_t code_load(uint32_t off, const unsigned char *base) {
    uint32_t t;
    if constexpr (ACROSS) asm volatile("; PQR %0\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(t) : "v"(off), "s"(base), "n"(4 * W) : "memory");
    else asm volatile("; PQI %0\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(t) : "v"(off), "s"(base), "n"(4 * W) : "memory");
    return t;
}

// Builds of the epilogue, for A/B runs: 0 = bounds-checked arithmetic, every wait for everything; 1 = + the short way
// for whole tiles; 2 (the library's) = + stores left in flight behind a counted wait, plain barrier; 3 = + hand-issued
// scale / code loads with one counted wait.  On one device all four give the same launch time within 0.3 %
// (profiles/r03_epilogue_ab.txt) although 3 shortens the epilogue itself by 0.7 us a tile: the launch runs at the clock
// the chip's power limit allows, and time taken out of the idle phases comes back as a lower clock.
#ifndef TL_EPI
#define TL_EPI 2
#endif

constexpr int TL_THREADS = 512;
constexpr int TL_WAVES = 8;
constexpr int TL_BK = 64;    // k per LDS tile
constexpr int TL_GROUP_T = 4; // token tiles per group of the tile -> block map

// byte offset of 16-byte slot `slot` (0..7) of row `row` inside a [rows][64] 16-bit tile
__device__ inline uint32_t xt_off(int row, int slot) { return (uint32_t)row * 128u + (uint32_t)((slot ^ (row & 7)) << 4); }

// WAVES_T of the 8 waves lie along the token axis, the rest along the row axis; a wave owns WN weight
// fragments x WT token fragments, so the block tile is (8/WAVES_T * WN * 16) rows x (WAVES_T * WT * 16)
// tokens.
// ABL (diagnostic builds only; 0 in the library): bit 0 skips the LDS store of the next tile and the
// barrier, bit 1 the activation loads, bit 2 the epilogue stores, bit 3 the gathers, bit 4 the activation
// fragment reads from LDS.
template <typename AT, typename CodeT, int D, int M, int WAVES_T, int WN, int WT, int KT, int ABL = 0, int NWV = TL_WAVES>
__global__ __launch_bounds__(NWV * 64) void gemm_tiled_kernel(
    const uint16_t *__restrict__ x, int T, int64_t x_stride, const u32x4 *__restrict__ packed,
    const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales, const uint16_t *__restrict__ bias,
    TileMap tmap, int G, int K, int nbits, int R, uint16_t *__restrict__ y, int64_t y_stride, int n_tiles,
    int t_tiles) {
    typedef typename Act<AT>::vec8 vec8;
    typedef StreamFmt<sizeof(CodeT), D, M> F;
    constexpr int CB = sizeof(CodeT);
    constexpr int EB = 2 * D;
    constexpr int CG = F::CG, KS = F::KS, CPS = F::CPS;
    constexpr int WAVES_N = NWV / WAVES_T;
    constexpr int BN = WAVES_N * WN * 16;            // weight rows per block
    constexpr int BT = WAVES_T * WT * 16;            // tokens per block
    constexpr int XT_BYTES = BT * TL_BK * 2;         // one activation tile
    constexpr int STAGE_BYTES = KT * XT_BYTES;       // one ring stage: KT consecutive tiles
    constexpr int NBUF = 3;                          // ring stages: one being read, two in flight
    constexpr int PPW = BT / (8 * NWV);              // one-KiB pieces per wave per tile
    // The codes of a chunk are consumed in PIECES register pieces (a half or a quarter of a lane's 16
    // bytes), so that the next ones can be prefetched with few live registers.
    constexpr int SPP = KS >= 8 ? KS / 4 : KS >= 4 ? KS / 2 : KS;   // k-steps per code piece
    constexpr int PIECES = KS / SPP;                 // pieces per chunk: 1, 2 or 4
    constexpr int PW = 4 / PIECES;                   // dwords per piece
    constexpr int TPP = SPP / 2;                     // 64-k tiles per piece
    // The main loop advances by U tiles: whole pieces and whole ring stages.
    constexpr int U = KT > TPP ? KT : TPP;
    constexpr int PPU = U / TPP;                     // pieces per loop step
    constexpr int SPU = U / KT;                      // ring stages per loop step
    typedef uint32_t piece_t __attribute__((ext_vector_type(PW)));
    // hand-pipelined LDS reads in the main loop: one codebook, fragments taken in pairs; d = 8, and the smaller
    // entries with one-byte codes (their images have 256 bytes per entry like d = 8's, so a gather address is the
    // code byte dropped into the lane's offset)
    constexpr bool PIPE = (D == 8 || CB == 1) && M == 1 && WN % 2 == 0 && WT % 2 == 0 && !(D < 8 && WT == 8) && !(ABL & (8 | 16));
    constexpr int GR = 2 * CPS;                      // LDS reads per pair of gathered fragments
    // the same format on the shapes with a single weight fragment per wave: hand-issued reads, in batches of k-steps
    constexpr bool BATCH = D == 8 && M == 1 && !PIPE && !(ABL & (8 | 16));
    // Blocks walk several tiles where the main loop is scheduled by hand; the compiler-scheduled formats keep one
    // tile per block (measured: their main loop comes out 6-9 % slower inside the tile loop).
    constexpr bool PERSIST = PIPE || BATCH;
    static_assert(KS % 2 == 0 && U % TPP == 0 && U % KT == 0, "pieces and stages must tile the loop step");
    static_assert(PPW >= 1, "tile too small for the block");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xt = smem + ST_MAX_LDS_CB;        // NBUF activation tiles behind the codebook image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int wn = wave / WAVES_T;                   // wave's position along n
    const int wt = wave % WAVES_T;                   // and along t
    // ABL bit 5 (diagnostic builds): `bias` is a buffer of 8 wall-clock stamps per block instead of a bias
    auto stamp = [&](int k) {
        if constexpr (ABL & 32) {
            if (tid == 0) {
                unsigned long long *st = reinterpret_cast<unsigned long long *>(const_cast<uint16_t *>(bias)) + (size_t)blockIdx.x * 8;
                st[k] = wall_clock64();
                if (k == 1 || k == 2) st[k + 4] = __builtin_amdgcn_s_memtime();   // shader clock around the main loop
            }
        }
    };
    stamp(0);

    // ---- XCD-aware, persistent tile assignment ---------------------------------------------------
    // The (row tile, token tile) pairs are cut into 8 contiguous runs, one per XCD (blocks b and b+8 share an
    // XCD and its L2).  Block b is slot b>>3 of XCD b&7 and takes pairs slot, slot + nslots, ... of that run: the
    // blocks an XCD runs at a time work on consecutive pairs, and a block keeps its codebook image, its ring
    // position and its place on the CU from one tile to the next.
    // Within a run the pairs go in groups of TL_GROUP_T token tiles: GROUP_T token tiles of one row tile, then of
    // the next row tile, ...  The ~32 blocks of an XCD then share 4 activation tiles and 8 row tiles' codes in its
    // L2, instead of 1 and 32: fewer L2 misses for the same work.
    const int total = n_tiles * t_tiles;
    const int per_xcd = (total + 7) >> 3;
    const int nslots = (int)(gridDim.x >> 3);
    const int run0 = (int)(blockIdx.x & 7) * per_xcd;
    int pin = (int)(blockIdx.x >> 3);                // position in the run
    if (pin >= per_xcd || run0 + pin >= total) return;
    struct TileSel { int part, row0, row_end, nfr, f0, t0; };
    auto tile_of = [&](int pair) {
        const int gwidth = TL_GROUP_T * n_tiles;
        const int gid = pair / gwidth, gin = pair - gid * gwidth;
        const int gfirst = gid * TL_GROUP_T;
        const int gsize = t_tiles - gfirst < TL_GROUP_T ? t_tiles - gfirst : TL_GROUP_T;
        const TileLoc loc = locate_tile(tmap, gin / gsize, BN);
        TileSel ts;
        ts.part = loc.part;
        ts.row0 = loc.row0;
        ts.row_end = loc.row0 + loc.nrows;
        ts.nfr = (loc.nrows + 15) >> 4;              // valid fragments in this row tile
        ts.f0 = loc.row0 >> 4;
        ts.t0 = (gfirst + gin % gsize) * BT;
        return ts;
    };
    TileSel cur_tile = tile_of(run0 + pin);
    const int Kin = G * D;
    const int nchunks = (G + CG - 1) / CG;

    // ---- activation tiles: LDS-DMA (global_load_lds), NBUF buffers, two tiles in flight ----------------
    // A tile is [BT][64] 16-bit values = BT/8 one-KiB pieces of 8 rows; a wave-instruction writes one piece
    // linearly (LDS address = piece base + lane*16), so the XOR swizzle goes on the per-lane SOURCE: LDS
    // slot l%8 of row l/8 receives logical 16-byte slot (l%8) ^ (l/8) of that row.  Tokens past T read the
    // last valid row (their outputs are never stored).
    const int ntiles_k = Kin / TL_BK;                // Kin % (64*U) == 0, checked on the host
    const int nstages = ntiles_k / KT;
    const int nsteps = ntiles_k / U;
    // what the request lambdas below read of the tile they work for: set by bind()
    const unsigned char *xbase;
    uint32_t goff[PPW];
    const unsigned char *fbase[WN];                  // fragment bases are wave-uniform (scalar registers)
    auto bind = [&](const TileSel &ts) {
        xbase = reinterpret_cast<const unsigned char *>(x + (size_t)ts.t0 * x_stride);
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            int row = (wave * PPW + j) * 8 + (lane >> 3);
            row = ts.t0 + row < T ? row : T - 1 - ts.t0;
            goff[j] = (uint32_t)row * (uint32_t)(x_stride * 2) + (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int fi = wn * WN + i;
            const int f = ts.f0 + (fi < ts.nfr ? fi : ts.nfr - 1);
            fbase[i] = reinterpret_cast<const unsigned char *>(packed) + (size_t)f * nchunks * 1024;
        }
    };
    bind(cur_tile);
    auto x_issue = [&](int st, int buf) {            // stage st -> ring slot buf (both wave-uniform)
        if constexpr (ABL & 2) return;
        // a stage past the end re-reads the last one (nobody reads that slot again): no branch around a load
        const uint32_t kb = (uint32_t)(st < nstages ? st : nstages - 1) * (KT * TL_BK * 2);
        unsigned char *dst = xt + buf * STAGE_BYTES + wave * (PPW * 1024);
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(xbase + (goff[j] + kb + it * (TL_BK * 2))),
                    (__attribute__((address_space(3))) void *)(dst + it * XT_BYTES + j * 1024), 16, 0, 0);
    };
    // one of the KT * PPW wave-instructions of x_issue: piece j of tile `it` of the stage
    auto x_issue_one = [&](int st, int buf, int it, int j) {
        if constexpr (ABL & 2) return;
        const uint32_t kb = (uint32_t)(st < nstages ? st : nstages - 1) * (KT * TL_BK * 2);
        unsigned char *dst = xt + buf * STAGE_BYTES + wave * (PPW * 1024);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(xbase + (goff[j] + kb + it * (TL_BK * 2))),
            (__attribute__((address_space(3))) void *)(dst + it * XT_BYTES + j * 1024), 16, 0, 0);
    };
    x_issue(0, 0);
    x_issue(1, 1);

    // The PPU code pieces of loop step u, requested with inline-asm loads: the compiler neither moves
    // them across the DMA issue nor waits for them on its own (it drains vmcnt to 0 for any tracked load
    // that is pending together with an LDS-DMA), so the counted waits below are exact.  Until `landed`
    // has been applied to a set after such a wait, its registers must not be read -- by the compiler either:
    // a register with a load in flight never lives across a branch, a join or a loop back edge (where a
    // copy could be inserted); tools/lint_inflight.py checks the generated code for both ("; PQI" marks a
    // hand-issued load or LDS read, "; PQL" the point from which its register may be read).
    auto load_pieces = [&](piece_t (&dst)[PPU][WN], int u, auto across_c) {
        constexpr bool ACROSS = decltype(across_c)::value;
#pragma unroll
        for (int pi = 0; pi < PPU; ++pi) {
            const int p = u * PPU + pi;
            const uint32_t off =
                (uint32_t)(p / PIECES) * 1024u + (uint32_t)lane * 16u + (uint32_t)(p % PIECES) * (16u / PIECES);
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                static_for<PW>([&](auto wc) { dst[pi][i][decltype(wc)::value] = code_load<decltype(wc)::value, ACROSS>(off, fbase[i]); });
            }
        }
    };
    // load k of load_pieces' PPU * WN * PW, in its order
    auto load_piece_one = [&](piece_t (&dst)[PPU][WN], int u, auto kc) {
        constexpr int k = decltype(kc)::value, pi = k / (WN * PW), i = (k / PW) % WN, w = k % PW;
        const int p = u * PPU + pi;
        const uint32_t off =
            (uint32_t)(p / PIECES) * 1024u + (uint32_t)lane * 16u + (uint32_t)(p % PIECES) * (16u / PIECES);
        const uint32_t t = code_load<w, false>(off, fbase[i]);
        dst[pi][i][w] = t;
    };
    constexpr int CL = PPU * WN * PW;                // code loads per loop step
    constexpr bool DEEP = KT > 1 && CL <= 8;         // prefetch codes two steps ahead (three register sets)
    static_assert(!DEEP || SPU == 1, "deep code prefetch assumes one stage per step");
    // makes every later read of the set depend on this point in the instruction stream
    auto landed = [&](piece_t (&c)[PPU][WN]) {
#pragma unroll
        for (int pi = 0; pi < PPU; ++pi) {
#pragma unroll
            for (int i = 0; i < WN; ++i) {
#pragma unroll
                for (int w = 0; w < PW; ++w) {
                    uint32_t t = c[pi][i][w];
                    asm volatile("; PQL %0" : "+v"(t));
                    c[pi][i][w] = t;
                }
            }
        }
    };
    // The codes of the first two steps travel while the codebook image is built (that code has loops): these
    // are ordinary, compiler-tracked loads.
    auto load_pieces_tracked = [&](piece_t (&dst)[PPU][WN], int u) {
#pragma unroll
        for (int pi = 0; pi < PPU; ++pi) {
            const int p = u * PPU + pi;
            const uint32_t off =
                (uint32_t)(p / PIECES) * 1024u + (uint32_t)lane * 16u + (uint32_t)(p % PIECES) * (16u / PIECES);
#pragma unroll
            for (int i = 0; i < WN; ++i) dst[pi][i] = *reinterpret_cast<const piece_t *>(fbase[i] + off);
        }
    };
    piece_t c0[PPU][WN], c1[PPU][WN], c2[PPU][WN];
    load_pieces_tracked(c0, 0);
    load_pieces_tracked(c1, nsteps > 1 ? 1 : 0);

    // ---- codebook image ---------------------------------------------------------------------------
    auto build_image = [&](int part, int tid_) {
        CodebookImage<D, NWV * 64> img;
        img.load(reinterpret_cast<const unsigned char *>(codebooks) + (size_t)part * M * K * EB, M * K, R, tid_);
        img.store(smem, M * K, R, tid_);
    };
    build_image(cur_tile.part, tid);
    int img_part = cur_tile.part;

    f32x4 acc[WN][WT];
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t shift = (uint32_t)__builtin_ctz((unsigned)(EB * R));
    const uint32_t lane_off = (uint32_t)(lane & (R - 1)) * EB;
    const uint32_t cb_stride = (uint32_t)K * EB * R;
    // LDS byte addresses for the hand-issued reads: this lane's copy of codebook entry 0, and its slot of
    // activation-tile row (first token fragment of the wave) for the two k-steps of a tile
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t img_lane = lds_base + lane_off;
    uint32_t xlane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        xlane[ks] = lds_base + ST_MAX_LDS_CB + xt_off(wt * WT * 16 + r16, ks * 4 + q);

    // everything issued so far (two tiles, the image) has landed before the first read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(c0);
    landed(c1);
    __syncthreads();
    stamp(1);

    // Loop step u of U tiles: request the codes of a later step, then per ring stage: issue the DMA of
    // stage st+2, compute stage st, wait until only what was issued after the DMA of stage st+1 is
    // outstanding (that stage has landed for this wave), barrier (landed for every wave; and everybody is
    // done reading stage st, whose slot the next DMA overwrites).  No branch around a load and no
    // __syncthreads() in the loop: either would drain vmcnt and stall the DMA pipeline.
    //   DEEP (several tiles per stage, small code sets): three code sets; step u requests the codes of step
    //     u+2, which are older than everything step u+1 issues and so have landed after the wait of step u+1;
    //   otherwise two sets: step u requests the codes of step u+1, which are older than the DMA issued in the
    //     step's last stage and so have landed after the step's last wait.
    int buf = 0, buf2 = 2;
    // `own` (DEEP only): the step also waits for its own request, so that no code load is in flight when it ends
    // -- used for the last step before a loop back edge or a branch.
    // Operand registers of the hand-pipelined stage.  With one stage per step they live across stages: a stage can
    // request the first operands of the next one (mode bit 1), which then finds them issued (mode bit 0); the
    // register set the gathers rotate through starts at `mode >> 2`.  Never across a branch or the loop's back edge.
    u32x4 xfr[2][WT];
    GatherFrag<D> P[3][2];
    constexpr int NGK = KT * 2 * (WN / 2);           // groups per stage of the hand-pipelined path
    constexpr bool XS = PIPE && D == 8 && SPU == 1 && CB == 1 && !(ABL & 1);   // two-byte codes: no registers to spare
    auto step = [&](const piece_t (&cur)[PPU][WN], piece_t (&lands)[PPU][WN], piece_t (&req)[PPU][WN], int u, auto own_c,
                    auto mode_c) {
        constexpr bool own = decltype(own_c)::value;
        constexpr int mode = decltype(mode_c)::value;
        constexpr bool PRE = XS && (mode & 1), NEXT = XS && (mode & 2);
        constexpr int BASE = XS ? (mode >> 2) % 3 : 0;
        constexpr int AHEAD = DEEP ? 2 : 1;
        const int ureq = u + AHEAD < nsteps ? u + AHEAD : nsteps - 1;
        if constexpr (!PIPE) load_pieces(req, ureq, std::integral_constant<bool, false>{});
        auto stage = [&](auto sgc) {     // sgc: the stage's number within the step, a compile-time constant for PIPE
            const int sg = sgc;
            if constexpr (!PIPE) x_issue(u * SPU + sg + 2, buf2);
            if constexpr (PIPE) {
                constexpr int sg = decltype(sgc)::value;
                // The stage's global-memory requests -- in a step's first stage the code loads, then the activation
                // DMA of stage st+2 -- go out one or two at a time behind the last MFMA of each group instead of in
                // one burst at the head of the stage; codes first, so that they are older than the stage's DMA.
                constexpr int NCODE = sg == 0 ? CL : 0, NOPS = NCODE + KT * PPW;
                auto vm_op = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < NCODE) load_piece_one(req, ureq, kc);
                    else x_issue_one(u * SPU + sg + 2, buf2, (k - NCODE) / PPW, (k - NCODE) % PPW);
                };
                // One stage = NKS k-steps x NPAIR fragment pairs = NG groups of 2 gathers + 2*WT MFMAs, software-
                // pipelined by hand: the gathers of group g+2 are issued before the MFMAs of group g (three
                // register sets), and the activation fragments of k-step kk+1 before the first MFMA of k-step kk
                // (two sets).  LDS reads return in order, so `lgkmcnt(n)` = all but the n newest.
                constexpr int NKS = KT * 2, NPAIR = WN / 2, NG = NKS * NPAIR;
                const uint32_t xs0 = xlane[0] + (uint32_t)buf * STAGE_BYTES, xs1 = xlane[1] + (uint32_t)buf * STAGE_BYTES;
                auto gather_pair = [&](auto gc, GatherFrag<D> (&dst)[2]) {        // group gc -> its two weight fragments
                    constexpr int g = decltype(gc)::value, kk = g / NPAIR, pp = g % NPAIR;
                    constexpr int tt_in = kk / 2, ks = kk % 2;
                    const int tt = sg * KT + tt_in;
                    const int sidx = (tt % TPP) * 2 + ks;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        uint32_t ad[CPS];
#pragma unroll
                        for (int cc = 0; cc < CPS; ++cc)
                            ad[cc] = gather_addr8<CB>(cur[tt / TPP][2 * pp + e], sidx * CPS + cc, (uint32_t)nbits, shift, img_lane);
                        dst[e].read(ad);
                    }
                };
                auto read_x = [&](auto kc) {                               // the WT activation fragments of k-step kk
                    constexpr int kk = decltype(kc)::value;
                    static_for<WT>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int off = (kk / 2) * XT_BYTES + j * 2048;
                        xfr[kk & 1][j] = lds_read16<off>((kk % 2) ? xs1 : xs0);
                    });
                };
                // reads group h issues before its MFMAs: the next k-step's activations (first group of a k-step)
                auto xr = [](int h) constexpr { return (h >= 0 && h % NPAIR == 0 && h / NPAIR + 1 < NKS) ? WT : 0; };
                static_assert(NG == NGK, "group count");
                if constexpr (!PRE) {
                    read_x(std::integral_constant<int, 0>{});
                    gather_pair(std::integral_constant<int, 0>{}, P[BASE % 3]);
                    if constexpr (NG > 1) gather_pair(std::integral_constant<int, (NG > 1 ? 1 : 0)>{}, P[(1 + BASE) % 3]);
                }
                static_for<NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value, kk = g / NPAIR, pp = g % NPAIR, c = (g + BASE) % 3, xb = kk & 1;
                    constexpr bool G2 = g + 2 < NG, XR = xr(g) > 0;
                    constexpr int g2 = G2 ? g + 2 : g, kk2 = g2 / NPAIR, pp2 = g2 % NPAIR;
                    constexpr int newer_all = pipe_newer<NG, NPAIR, NKS, WT, PRE, GR>(g);
                    constexpr int newer = newer_all > 15 ? 15 : newer_all;   // the counter has four bits; waiting for more is safe
                    constexpr bool HANDOVER = NEXT && g == NG - 1;     // this group's gaps carry the next stage's first reads
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(newer));
                    P[c][0].landed();
                    P[c][1].landed();
                    if constexpr (pp == 0) {
#pragma unroll
                        for (int j = 0; j < WT; ++j) asm volatile("; PQL %0" : "+v"(xfr[xb][j]));
                    }
                    uint32_t na[4] = {0u, 0u, 0u, 0u};
                    if constexpr (HANDOVER) {
                        // Every LDS read of this stage has returned (the wait above was lgkmcnt(0)): the stage's end
                        // -- next stage's DMA landed for this wave, then for all, and nobody reads this stage's slot
                        // any more -- comes here, in front of the group's MFMAs, which then cover the barrier and
                        // the latency of the next stage's first operands.
                        static_assert(pipe_newer<NG, NPAIR, NKS, WT, PRE>(NG - 1) == 0, "last group waits for everything");
                        if ((DEEP && !own) || SPU > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * PPW + CL) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * PPW) : "memory");
                        landed(lands);
                        if constexpr (DEEP && own) landed(req);
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    // the group's 2*WT MFMAs with its LDS traffic spread over the gaps between them, one or two
                    // instructions a gap: both gather addresses, the two gathers, then the next k-step's activations
                    uint32_t ga[2][CPS];
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                        for (int cc = 0; cc < CPS; ++cc) ga[e2][cc] = 0u;
                    const u32x4 pa[2] = {P[c][0].value(), P[c][1].value()};
                    static_for<2 * WT>([&](auto mc) {
                        constexpr int m = decltype(mc)::value, j = m / 2, e = m % 2;
                        acc[2 * pp + e][j] = Act<AT>::mfma16(__builtin_bit_cast(vec8, pa[e]),
                                                             __builtin_bit_cast(vec8, xfr[xb][j]), acc[2 * pp + e][j]);
                        if constexpr (G2 && m == 0) {
                            const int tt = sg * KT + kk2 / 2;
                            const int sidx = (tt % TPP) * 2 + kk2 % 2;
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                                for (int cc = 0; cc < CPS; ++cc)
                                    ga[e2][cc] = gather_addr8<CB>(cur[tt / TPP][2 * pp2 + e2], sidx * CPS + cc, (uint32_t)nbits, shift, img_lane);
                        }
                        if constexpr (G2 && (m == 1 || m == 2)) P[(g2 + BASE) % 3][m - 1].read(ga[m - 1]);
                        if constexpr (HANDOVER) {
                            // fillers of the hand-over: addresses, gathers of the next stage's groups 0 and 1 (from the
                            // codes that have just landed), its first activation fragments (next ring slot)
                            constexpr int NB_ = (BASE + NG) % 3;
                            constexpr int last = 2 * WT - 1;
                            auto filler = [&](auto ic) {
                                constexpr int i = decltype(ic)::value;
                                if constexpr (i == 0) {
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        const int kkh = h / NPAIR, pph = h % NPAIR;
                                        const int tth = kkh / 2, sidxh = (tth % TPP) * 2 + kkh % 2;
#pragma unroll
                                        for (int e2 = 0; e2 < 2; ++e2)
                                            na[2 * h + e2] = gather_addr8<CB>(lands[tth / TPP][2 * pph + e2], sidxh, (uint32_t)nbits, shift, img_lane);
                                    }
                                } else if constexpr (i <= 4) {
                                    if constexpr (D == 8) {
                                        const uint32_t one[1] = {na[i - 1]};
                                        P[((i - 1) / 2 + NB_) % 3][(i - 1) % 2].read(one);
                                    }
                                } else {
                                    constexpr int jx = i - 5;
                                    const uint32_t nbuf = buf == NBUF - 1 ? 0u : (uint32_t)buf + 1u;
                                    xfr[0][jx] = lds_read16<jx * 2048>(xlane[0] + nbuf * STAGE_BYTES);
                                }
                            };
                            constexpr int NFILL = 5 + WT;
                            if constexpr (m < last) {
                                if constexpr (m < NFILL) filler(std::integral_constant<int, (m < NFILL ? m : 0)>{});
                            } else {
                                static_for<(NFILL > last ? NFILL - last : 0)>([&](auto dc) {
                                    filler(std::integral_constant<int, last + decltype(dc)::value>{});
                                });
                            }
                        }
                        if constexpr (XR && m >= 3 && m - 3 < WT) {
                            constexpr int jx = m - 3, kx = XR ? kk + 1 : kk;
                            constexpr int off = (kx / 2) * XT_BYTES + jx * 2048;
                            xfr[kx & 1][jx] = lds_read16<off>((kx % 2) ? xs1 : xs0);
                        }
                        if constexpr (m == 2 * WT - 1 && !HANDOVER) {
                            // with a hand-over the stage's requests all go out before its last group (they are counted
                            // by the wait in front of it)
                            constexpr int NGV = NEXT ? NG - 1 : NG;
                            constexpr int k0 = g * NOPS / NGV, k1 = (g + 1) * NOPS / NGV;
                            static_for<k1 - k0>([&](auto dc) { vm_op(std::integral_constant<int, k0 + decltype(dc)::value>{}); });
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    static_assert(3 + WT <= 2 * WT, "not enough MFMA gaps for the group's reads");
                });
            } else if constexpr (BATCH) {
                // Few MFMAs per k-step (small tiles): the operands of NB k-steps are requested in one go and
                // consumed after a single wait, one batch ahead of the MFMAs (two register sets), instead of
                // one LDS round trip per k-step.
                constexpr int NKS = KT * 2;
                constexpr int REGS_PER_KS = 4 * (WT + WN);
                constexpr int NB0 = 48 / REGS_PER_KS;
                constexpr int NB = NB0 >= NKS ? NKS : NB0 >= 4 ? 4 : NB0 >= 2 ? 2 : 1;
                constexpr int NBATCH = NKS / NB;
                const uint32_t xs0 = xlane[0] + (uint32_t)buf * STAGE_BYTES, xs1 = xlane[1] + (uint32_t)buf * STAGE_BYTES;
                u32x4 xr[2][NB][WT], ar[2][NB][WN];
                auto issue = [&](auto bc) {
                    constexpr int b = decltype(bc)::value, slot = b & 1;
                    static_for<NB>([&](auto kbc) {
                        constexpr int kb = decltype(kbc)::value, kk = b * NB + kb, it = kk / 2, ks = kk % 2;
                        const int tt = sg * KT + it;
                        const int sidx = (tt % TPP) * 2 + ks;
                        static_for<WT>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            xr[slot][kb][j] = lds_read16<it * XT_BYTES + j * 2048>(ks ? xs1 : xs0);
                        });
#pragma unroll
                        for (int i = 0; i < WN; ++i) {
                            ar[slot][kb][i] = lds_read16<0>(gather_addr8<CB>(cur[tt / TPP][i], sidx, (uint32_t)nbits, shift, img_lane));
                        }
                    });
                };
                issue(std::integral_constant<int, 0>{});
                static_for<NBATCH>([&](auto bc) {
                    constexpr int b = decltype(bc)::value, slot = b & 1;
                    constexpr bool more = b + 1 < NBATCH;
                    if constexpr (more) issue(std::integral_constant<int, more ? b + 1 : b>{});
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(more ? NB * (WT + WN) : 0));
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
                        for (int j = 0; j < WT; ++j) asm volatile("; PQL %0" : "+v"(xr[slot][kb][j]));
#pragma unroll
                        for (int i = 0; i < WN; ++i) asm volatile("; PQL %0" : "+v"(ar[slot][kb][i]));
                    }
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                        for (int i = 0; i < WN; ++i)
#pragma unroll
                            for (int j = 0; j < WT; ++j)
                                acc[i][j] = Act<AT>::mfma16(__builtin_bit_cast(vec8, ar[slot][kb][i]),
                                                            __builtin_bit_cast(vec8, xr[slot][kb][j]), acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
            // the lane's row and k-quad from an opaque copy of the lane number: the activation addresses derived from
            // them are then worked out per stage (a few instructions) instead of staying in registers across the
            // whole main loop for the code behind it
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));
            const int r16 = lane_s & 15, q = lane_s >> 4;
#pragma unroll
            for (int it = 0; it < KT; ++it) {
                const int tt = sg * KT + it;           // tile within the loop step (compile-time)
                const unsigned char *xb = xt + buf * STAGE_BYTES + it * XT_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int s = (tt % TPP) * 2 + ks; // k-step within the piece
                    vec8 xf[WT];
#pragma unroll
                    for (int j = 0; j < WT; ++j) {
                        const int row = (wt * WT + j) * 16 + r16;
                        if constexpr (ABL & 16) {
                            const u32x4 fake{(uint32_t)row * 0x3f1u | 0x3f003f00u, 0x3f803f80u, (uint32_t)lane | 0x3f003f00u, 0x40004000u};
                            xf[j] = __builtin_bit_cast(vec8, fake);
                        } else {
                            xf[j] = *reinterpret_cast<const vec8 *>(xb + xt_off(row, ks * 4 + q));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < WN; ++i) {
                        vec8 a;
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            uint32_t idx[CPS];
#pragma unroll
                            for (int c = 0; c < CPS; ++c)
                                idx[c] = code_at<CB>(cur[tt / TPP][i], (s * CPS + c) * M + m, (uint32_t)nbits);
                            vec8 am;
                            if constexpr (ABL & 8) {
                                const u32x4 fake{idx[0], idx[0] + 1, idx[0] ^ 5, idx[0] * 3};
                                am = __builtin_bit_cast(vec8, fake);
                            } else {
                                am = gather_frag<AT, D>(smem + m * cb_stride, idx, shift, lane_off);
                            }
                            a = m == 0 ? am : Act<AT>::add8(a, am);
                        }
#pragma unroll
                        for (int j = 0; j < WT; ++j) acc[i][j] = Act<AT>::mfma16(a, xf[j], acc[i][j]);
                    }
                }
            }
            }
            if constexpr (NEXT) {
                // the stage's end was taken in front of its last group (hand-over)
            } else if constexpr (!(ABL & 1)) {
                // younger than the DMA of stage st+1: this stage's DMA and, in a step's first stage, its code loads
                if (sg == 0 && ((DEEP && !own) || SPU > 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * PPW + CL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * PPW) : "memory");
                if (sg == SPU - 1) {
                    landed(lands);
                    if constexpr (DEEP && own) landed(req);
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                if (sg == SPU - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    landed(lands);
                    if constexpr (DEEP && own) landed(req);
                }
            }
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            buf2 = buf2 == NBUF - 1 ? 0 : buf2 + 1;
        };
        if constexpr (PIPE) {
            static_for<SPU>(stage);
        } else {
#pragma unroll
            for (int sg = 0; sg < SPU; ++sg) stage(sg);
        }
    };
    constexpr std::integral_constant<bool, false> keep{};   // the request stays in flight into the next step
    constexpr std::integral_constant<bool, true> land{};    // nothing in flight when the step ends
    for (int first = 1;; first = 0) {
        // step modes: bit 0 = the step before has requested this one's first operands, bit 1 = this one requests the
        // next one's, >> 2 = first gather register set.  Operands in flight only between steps with no branch between.
        constexpr std::integral_constant<int, 0> alone{};
        constexpr std::integral_constant<int, 2> opens{};
        constexpr std::integral_constant<int, 1 | 2 | ((NGK % 3) << 2)> middle{};
        constexpr std::integral_constant<int, 1 | (((2 * NGK) % 3) << 2)> last3{};
        constexpr std::integral_constant<int, 1 | ((NGK % 3) << 2)> last2{};
        if constexpr (DEEP) {
            int u = 0;
            for (; u + 3 <= nsteps; u += 3) {
                step(c0, c1, c2, u, keep, opens);
                step(c1, c2, c0, u + 1, keep, middle);
                step(c2, c0, c1, u + 2, land, last3);
            }
            if (u < nsteps) {
                step(c0, c1, c2, u, land, alone);
                if (u + 1 < nsteps) step(c1, c2, c0, u + 1, land, alone);
            }
        } else {
            int u = 0;
            for (; u + 2 <= nsteps; u += 2) {
                step(c0, c1, c1, u, land, opens);
                step(c1, c0, c0, u + 1, land, last2);
            }
            if (u < nsteps) step(c0, c1, c1, u, land, alone);
        }
        // Every step above ended with its code loads landed and with a barrier: no wave reads the ring or gathers
        // any more.  The activation DMA of the two stages past the end (re-reads of the last stage) may still be
        // writing the ring; a wave's DMA lands in issue order and writes that wave's pieces only.
        if (first) stamp(2);

        // ---- the block's next tile: its first two stages and its first codes travel during this tile's epilogue ----
        const int npin = pin + nslots;
        const bool more = PERSIST && npin < per_xcd && run0 + npin < total;
        const TileSel done = cur_tile;
        pin = npin;

        // ---- epilogue: scale, bias, round, store ---------------------------------------------------------------
        // A lane holds 4 consecutive rows (8 bytes of y) of one token per fragment.  Fragments are taken in
        // pairs: v_permlane16_swap trades the odd 16-lane rows of one register with the even rows of the
        // other, after which every lane owns 8 consecutive rows of one token (k-quad q even: rows 4q..4q+7 of
        // the pair's first fragment; q odd: rows 4(q-1)..+7 of the second), stored as 16 bytes -- half the
        // store instructions, 64 contiguous bytes per token and instruction instead of 32 (y is 16-byte aligned
        // with rows a multiple of 16 bytes apart: checked on the host).
        // The scales (and bias) of the lane's four rows of every fragment are 8 contiguous bytes: one load each, all
        // requested before the first is used (one memory round trip for the whole epilogue).  Rows of a fragment are
        // whole or absent (partition sizes are multiples of 16); an absent fragment reads the partition's last rows.
        u32x2 sraw[WN], braw[WN];
        const bool has_bias = bias && !(ABL & 32);
        if constexpr (PERSIST && TL_EPI >= 3) {
            // Hand-issued, like the main loop's loads, so that one counted wait separates what the epilogue needs now
            // (scales, bias: usually in the L2) from what only the next tile needs (its first codes, often from HBM,
            // and its first two stages), which then travel behind the epilogue's arithmetic and stores.  No branch
            // between the requests and their waits: without a bias its loads re-read the scales.
            const uint16_t *bsrc = has_bias ? bias : scales;
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const int nb = done.row0 + (wn * WN + i) * 16 + q * 4;
                const int nc = nb + 4 <= done.row_end ? nb : done.row_end - 4;
                asm volatile("; PQI %0\n\tglobal_load_dwordx2 %0, %1, off" : "=v"(sraw[i]) : "v"(scales + nc) : "memory");
                asm volatile("; PQI %0\n\tglobal_load_dwordx2 %0, %1, off" : "=v"(braw[i]) : "v"(bsrc + nc) : "memory");
            }
            // the next tile, worked out while the requests above travel (past the last tile: this one again)
            cur_tile = tile_of(run0 + (more ? npin : npin - nslots));
            bind(cur_tile);
            load_pieces(c0, 0, std::integral_constant<bool, true>{});
            load_pieces(c1, nsteps > 1 ? 1 : 0, std::integral_constant<bool, true>{});
            // The ring starts over at slot 0 for every tile, so that the slot of every stage of the unrolled main loop
            // is a compile-time constant.  The DMA of the two stages past the end of the finished tile may still be
            // writing the ring; it comes from the same waves as what is issued here and lands first.
            buf = 0;
            buf2 = 2;
            x_issue(0, 0);
            x_issue(1, 1);
            static_assert(2 * CL + 2 * KT * PPW < 64, "the wait's count has six bits");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CL + 2 * KT * PPW) : "memory");
#pragma unroll
            for (int i = 0; i < WN; ++i) asm volatile("; PQL %0\n\t; PQL %1" : "+v"(sraw[i]), "+v"(braw[i]));
        } else {
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const int nb = done.row0 + (wn * WN + i) * 16 + q * 4;
                const int nc = nb + 4 <= done.row_end ? nb : done.row_end - 4;
                sraw[i] = *reinterpret_cast<const u32x2 *>(scales + nc);
                braw[i] = has_bias ? *reinterpret_cast<const u32x2 *>(bias + nc) : u32x2{0u, 0u};
            }
            if constexpr (PERSIST) {
                cur_tile = tile_of(run0 + (more ? npin : npin - nslots));
                bind(cur_tile);
                load_pieces_tracked(c0, 0);
                load_pieces_tracked(c1, nsteps > 1 ? 1 : 0);
            }
            // These loads are waited for here, before the DMA goes out: the compiler waits for a tracked load that is
            // pending beside an LDS-DMA with vmcnt(0), which would put the DMA's latency in front of the epilogue
            // arithmetic instead of behind it.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < WN; ++i) asm volatile("" : "+v"(sraw[i]), "+v"(braw[i]));
            if constexpr (PERSIST) {
                landed(c0);
                landed(c1);
                buf = 0;
                buf2 = 2;
                x_issue(0, 0);
                x_issue(1, 1);
            }
        }
        if (first) stamp(7);

        auto finish = [&](int i, int j, uint32_t &lo, uint32_t &hi) {   // fragment (i, j) -> two packed dwords
            uint16_t h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = Act<AT>::to_f32((uint16_t)(sraw[i][r >> 1] >> ((r & 1) * 16)));
                const float bi = (bias && !(ABL & 32)) ? Act<AT>::to_f32((uint16_t)(braw[i][r >> 1] >> ((r & 1) * 16))) : 0.f;
                h[r] = Act<AT>::from_f32(acc[i][j][r] * sc + bi);
                acc[i][j][r] = 0.f;
            }
            lo = h[0] | ((uint32_t)h[1] << 16);
            hi = h[2] | ((uint32_t)h[3] << 16);
        };
        // A wave whose part of the tile lies inside the matrix on both axes (all of them, except along a ragged edge)
        // takes the short way: no bounds checks, so its stores are exactly WN / 2 * WT and the wait behind them can
        // leave them in flight; scales applied to pairs of values (v_pk_mul_f32), the bias only where there is one;
        // one store address per wave, the fragments at constant distances from it.
        const bool whole = TL_EPI >= 1 && WN % 2 == 0 && done.t0 + (wt * WT + WT) * 16 <= T &&
                           done.row0 + (wn * WN + WN) * 16 <= done.row_end;
        auto finish_whole = [&](auto hb_c) {
            constexpr bool HB = decltype(hb_c)::value;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            auto pair_of = [](uint32_t w) { return f32x2{Act<AT>::to_f32((uint16_t)w), Act<AT>::to_f32((uint16_t)(w >> 16))}; };
            uint16_t *yp0 = y + (size_t)(done.t0 + wt * WT * 16 + r16) * y_stride + done.row0 + wn * WN * 16 +
                            ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);
#pragma unroll
            for (int ip = 0; ip < WN / 2; ++ip) {
                f32x2 sc[2][2], bi[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sc[e][h] = pair_of(sraw[2 * ip + e][h]);
                        bi[e][h] = HB ? pair_of(braw[2 * ip + e][h]) : f32x2{0.f, 0.f};
                    }
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    uint32_t w[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 &a = acc[2 * ip + e][j];
                            f32x2 v = f32x2{a[2 * h], a[2 * h + 1]} * sc[e][h];
                            if constexpr (HB) v = v + bi[e][h];
                            w[e][h] = Act<AT>::pack2(v.x, v.y);
                            a[2 * h] = 0.f;
                            a[2 * h + 1] = 0.f;
                        }
                    const auto sl = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
                    const auto sh = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
                    const u32x4 o4{sl[0], sh[0], sl[1], sh[1]};
                    if constexpr (ABL & 4) {
                        if (o4.x == 0x12344321u && o4.y == 0x43211234u && o4.z == 0x11111111u) yp0[0] = (uint16_t)o4.w;
                    } else {
                        *reinterpret_cast<u32x4 *>(yp0 + (size_t)j * 16 * y_stride + ip * 32) = o4;
                    }
                }
            }
        };
        if (whole) {
            if (has_bias) finish_whole(std::integral_constant<bool, true>{});
            else finish_whole(std::integral_constant<bool, false>{});
        } else if constexpr (WN % 2 == 0) {
#pragma unroll
            for (int ip = 0; ip < WN / 2; ++ip) {
                const int n0 = done.row0 + (wn * WN + 2 * ip) * 16;
                const int ns = n0 + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);   // first of this lane's 8 rows after the swap
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    uint32_t alo, ahi, blo, bhi;
                    finish(2 * ip, j, alo, ahi);
                    finish(2 * ip + 1, j, blo, bhi);
                    const auto sl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                    const auto sh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                    const int t = done.t0 + (wt * WT + j) * 16 + r16;
                    if (t >= T || ns >= done.row_end) continue;     // fragments are whole or absent (rows % 16 == 0)
                    uint16_t *yp = y + (size_t)t * y_stride + ns;
                    const u32x4 o4{sl[0], sh[0], sl[1], sh[1]};
                    if constexpr (ABL & 4) {
                        if (o4.x == 0x12344321u && o4.y == 0x43211234u && o4.z == 0x11111111u) yp[0] = (uint16_t)o4.w;
                        continue;
                    }
                    *reinterpret_cast<u32x4 *>(yp) = o4;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const int nb = done.row0 + (wn * WN + i) * 16 + q * 4;
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    uint32_t lo, hi;
                    finish(i, j, lo, hi);
                    const u32x2 o2{lo, hi};
                    const int t = done.t0 + (wt * WT + j) * 16 + r16;
                    if (t >= T || nb >= done.row_end) continue;
                    uint16_t *yp = y + (size_t)t * y_stride + nb;
                    if constexpr (ABL & 4) {
                        if (o2.x == 0x12344321u && o2.y == 0x43211234u) yp[0] = 1;   // keeps acc live
                        continue;
                    }
                    *reinterpret_cast<u32x2 *>(yp) = o2;
                }
            }
        }
        if (first) stamp(3);
        // The next tile's first stages have landed for this wave.  The DMA went out before the tile's stores and memory
        // operations retire in issue order: a wave that took the short way leaves its WN / 2 * WT stores in flight
        // (they are older than anything the next tile waits for by count), any other waits for everything.
        if (TL_EPI >= 2 && whole && !(ABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN / 2 * WT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PERSIST) {
            landed(c0);
            landed(c1);
        }
        if (!more) break;
        if (cur_tile.part != img_part) {             // block-uniform; every wave is past its last gather
            int tid_here = tid;                      // opaque: keeps the image's address arithmetic out of the loop's registers
            asm volatile("" : "+v"(tid_here));
            build_image(cur_tile.part, tid_here);
            img_part = cur_tile.part;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // (a plain barrier: __syncthreads() would wait for the stores)
        if constexpr (TL_EPI >= 2) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    }
    stamp(4);
}

// ---- per-dtype launcher ---------------------------------------------------------------------------
// (row_lo, row_hi: the launch covers these rows of the layer only -- the whole layer by default)
template <typename AT, typename CodeT, int D, int M, int WAVES_T, int WN, int WT, int KT>
static hipError_t tiled_launch_cfg(const GemmArgs &a, hipStream_t stream, int row_lo = 0, int row_hi = 0x7fffffff) {
    constexpr int BN = (TL_WAVES / WAVES_T) * WN * 16;
    constexpr int BT = WAVES_T * WT * 16;
    // d = 2 with one-byte codes: 64 copies (a copy per lane; 256 bytes per entry, which the hand-pipelined stage's
    // gather addresses assume) -- 64 KiB at 256 entries
    const int R = (D == 2 && sizeof(CodeT) == 1) ? 64 : stream_replication<D>(a.K, a.M);
    const int nbits = __builtin_ctz((unsigned)a.K);
    const int n_tiles = tiles_in_range(a.parts, BN, row_lo, row_hi);
    const int t_tiles = (a.T + BT - 1) / BT;
    const int total = n_tiles * t_tiles;
    if (total == 0) return hipSuccess;
    const int per_xcd = (total + 7) / 8;
    const size_t lds = (size_t)ST_MAX_LDS_CB + 3 * (size_t)KT * BT * TL_BK * 2;
    auto kern = gemm_tiled_kernel<AT, CodeT, D, M, WAVES_T, WN, WT, KT>;
    // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    // one block per CU (a block takes the whole LDS), at most: blocks walk their XCD's run of tiles
    static std::atomic<int> slots_cached[64];
    int nslots = slots_cached[dev].load(std::memory_order_acquire);
    if (nslots == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        nslots = cus / 8;
        slots_cached[dev].store(nslots, std::memory_order_release);
    }
    if (knobs().tiled_slots > 0) nslots = knobs().tiled_slots;   // tests and diagnosis: blocks per XCD (large: one tile per block)
    constexpr bool persist = D == 8 && M == 1;     // the kernel's PERSIST
    if (!persist || nslots > per_xcd) nslots = per_xcd;
    hipLaunchKernelGGL(kern, dim3(nslots * 8), dim3(TL_THREADS), lds, stream, static_cast<const uint16_t *>(a.x), a.T,
                       a.x_stride, static_cast<const u32x4 *>(a.codes), static_cast<const uint16_t *>(a.codebooks),
                       static_cast<const uint16_t *>(a.scales), static_cast<const uint16_t *>(a.bias),
                       make_tile_map_range(a.parts, BN, row_lo, row_hi), a.G, a.K, nbits, R, static_cast<uint16_t *>(a.y),
                       a.y_stride, n_tiles, t_tiles);
    return hipGetLastError();
}

// Tile shapes: {rows, tokens}.  Bigger tiles reuse each gathered weight fragment and
// each activation fragment more often; smaller ones give the 256 CUs enough blocks.
// kt = 64-k tiles per ring stage (256 / bt: every shape keeps 32 KiB a stage); Kin must be a multiple of 64 kt.
struct TiledCfg { int bn, bt, kt; };
static const TiledCfg kTiledCfgs[] = {{256, 256, 1}, {256, 128, 2}, {128, 128, 2}, {256, 64, 4}, {128, 64, 4}, {64, 64, 4},
                                      {512, 128, 2}};
constexpr int kNumTiledCfgs = 7;

// 64-k tiles per main-loop step of a shape with kt tiles per ring stage: whole stages and whole code pieces
static int tiled_step_tiles(const GemmArgs &a, int kt) {
    const int ks = ((16 / a.code_bytes) / a.M) * a.d / 8;             // k-steps per code chunk
    const int tpp = (ks >= 8 ? ks / 4 : ks >= 4 ? ks / 2 : ks) / 2;  // tiles per code piece
    return kt > tpp ? kt : tpp;
}

// Shape i for this format.  The formats whose weight fragment costs several LDS reads or a sum (d = 4, d = 2, two
// codebooks) run a 256 x 256 tile whose waves take 4 weight fragments x 8 token fragments (instead of 8 x 4) in the
// first slot, so that a fragment feeds 8 MFMAs, on the compiler-scheduled main loop (hand-pipelined, its two sets of
// 8 activation fragments do not fit the register file), and a 128 x 256 tile of 2 x 8 fragments a wave in the
// 128 x 128 shape's slot.
static TiledCfg tiled_cfg(const GemmArgs &a, int i) {
    if (i == 0 && a.M == 2) return TiledCfg{256, 256, 1};
    if (i == 2 && (a.M == 2 || a.d < 8)) return TiledCfg{128, 256, 1};
    if (i == 0 && (a.d == 4 || a.d == 2)) return TiledCfg{256, 256, 1};
    // the 512 x 128 shape exists for d = 8, one codebook, one-byte codes; other formats run 256 x 128 in its slot
    if (i == 6 && !(a.d == 8 && a.M == 1 && a.code_bytes == 1)) return kTiledCfgs[1];
    return kTiledCfgs[i];
}

constexpr double kSpeed512x128 = 1.355, kSpeed256x128 = 1.24;   // d = 8, one codebook, one-byte codes

static int choose_tiled_cfg(const GemmArgs &a, double *modelled = nullptr) {
    if (modelled) *modelled = -1.0;
    if (knobs().tiled_cfg >= 0) {
        const int v = knobs().tiled_cfg;
        if (v >= 0 && v < kNumTiledCfgs && (a.G * a.d) % (64 * tiled_step_tiles(a, tiled_cfg(a, v).kt)) == 0) return v;
    }
    // Shortest modelled launch: the blocks run in rounds of one per CU (every shape takes the whole LDS), a
    // block costs its tile area over the shape's measured throughput (PFLOP/s on a chip-filling problem:
    // bigger tiles reuse each gathered weight fragment and each activation fragment more often).  Padding
    // and half-filled last rounds are paid in full; ties go to the larger tile.
    // (256 x 128 re-measured over several rounds of blocks on gate_up_proj: 49.4 us a round against 90 for 512 x 128)
    static const double speed_one[kNumTiledCfgs] = {1.33, kSpeed256x128, 1.085, 1.13, 0.856, 0.622, kSpeed512x128};
    // two codebooks: every weight fragment is two gathers and eight rounded sums, so what counts is how many MFMAs a
    // fragment feeds -- 8 in the slots of the 256 x 256 and 128 x 128 shapes (256 x 256 and 128 x 256 tiles whose
    // waves take 8 token fragments), 4 or 2 elsewhere
    static const double speed_two[kNumTiledCfgs] = {0.92, 0.59, 0.87, 0.55, 0.50, 0.31, 0.59};
    // d = 4 and d = 2 (one- and two-byte codes): several narrow gathers per weight fragment, from an image with few
    // copies when the codes have two bytes -- again the shapes whose waves take 8 token fragments (the slots of
    // 256 x 256 and 128 x 128) are the fast ones.  Measured on an 8192 x 8192 layer at 4096 tokens.
    static const double speed_d4[kNumTiledCfgs] = {1.33, 1.30, 1.13, 1.22, 0.80, 0.56, 1.30};
    static const double speed_d2[kNumTiledCfgs] = {1.17, 0.85, 0.98, 0.81, 0.62, 0.40, 0.85};
    static const double speed_d4w[kNumTiledCfgs] = {1.28, 0.95, 1.09, 0.87, 0.66, 0.42, 0.95};
    static const double speed_d2w[kNumTiledCfgs] = {0.87, 0.49, 0.73, 0.48, 0.44, 0.24, 0.49};
    const bool per_slot = a.M == 2 || a.d < 8;
    const double *speed = a.M == 2 ? speed_two
                          : a.d == 4 ? (a.code_bytes == 1 ? speed_d4 : speed_d4w)
                          : a.d == 2 ? (a.code_bytes == 1 ? speed_d2 : speed_d2w) : speed_one;
    static const int order[kNumTiledCfgs] = {6, 0, 1, 2, 3, 4, 5};
    int best = 0;   // shape 0 takes any Kin the kernel supports
    double best_cost = -1.0;
    for (int i : order) {
        const TiledCfg c = tiled_cfg(a, i);
        if ((a.G * a.d) % (64 * tiled_step_tiles(a, c.kt))) continue;
        const long blocks = (long)tiles_in(a.parts, c.bn) * ((a.T + c.bt - 1) / c.bt);
        // (the two-codebook table is per slot; a one-codebook format whose slot holds another shape runs 256 x 128 there)
        const double sp = (per_slot || (c.bn == kTiledCfgs[i].bn && c.bt == kTiledCfgs[i].bt)) ? speed[i] : speed[1];
        const double cost = (double)((blocks + 255) / 256) * c.bn * c.bt / sp;
        if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = i; }
    }
    if (modelled) *modelled = best_cost;
    return best;
}

// A last partial round.  The blocks run in rounds of one per CU, so 896 tiles of 512 x 128 (gate_up_proj of Llama-3-8B at
// 2048 tokens) cost four rounds although they are three and a half.  Where the model says it pays, the layer is cut along
// its rows: the first rows, as many 512-row tiles as fill whole rounds, run as one launch of 512 x 128 tiles, the rest as a
// second launch of 256 x 128 tiles (half the height: the half round becomes a whole one).  Measured on that layer against
// the best single launch, weights streaming from HBM (tools/ab_split.sh): 339 -> 330 us at 2048 tokens, 427 -> 409 at
// 2560, 517 -> 496 at 3072; nothing to gain where the tiles fill whole rounds (4096 tokens: 1792 tiles = 7 rounds).
// Returns the first row of the second launch, 0 for no split.
static int choose_tiled_split(const GemmArgs &a) {
    if (!(a.d == 8 && a.M == 1 && a.code_bytes == 1)) return 0;
    const int forced = knobs().tiled_split;
    if (forced == 0 || knobs().tiled_cfg >= 0) return 0;
    if ((a.G * a.d) % (64 * 2)) return 0;                       // both shapes walk K in steps of two 64-k tiles
    const int rtot = tiles_in(a.parts, 512);
    const long t_tiles = (a.T + 127) / 128;
    int r1;
    if (forced > 0) {
        r1 = forced;
    } else {
        const long full = rtot * t_tiles / 256;                 // whole rounds of 512 x 128 tiles
        if (full < 1) return 0;
        r1 = (int)(256 * full / t_tiles);
    }
    if (r1 < 1 || r1 >= rtot) return 0;
    // the end row of the r1-th 512-row tile, in the map's order (every partition tiled separately)
    int row = 0, left = r1, split = 0;
    for (int p = 0; p < a.parts.num && left > 0; ++p) {
        const int nt = (a.parts.size[p] + 511) / 512;
        if (left >= nt) { left -= nt; row += a.parts.size[p]; split = row; }
        else { split = row + left * 512; left = 0; }
    }
    if (split <= 0 || split >= a.N) return 0;
    if (forced > 0) return split;
    // rounds x tile area / measured throughput of the shape, plus what a second launch costs (12 us at 4096 inputs)
    const double c6 = 512.0 * 128.0 / kSpeed512x128, c1 = 256.0 * 128.0 / kSpeed256x128, launch = 7000.0 * 4096.0 / (double)(a.G * a.d);
    const long n1 = (long)r1 * t_tiles, n2 = (long)tiles_in_range(a.parts, 256, split, 0x7fffffff) * t_tiles;
    double single = -1.0;
    choose_tiled_cfg(a, &single);                               // the best single launch of any shape
    if (single < 0.0) return 0;
    const double two = (double)((n1 + 255) / 256) * c6 + (double)((n2 + 255) / 256) * c1 + launch;
    return two < 0.98 * single ? split : 0;
}

template <typename AT, typename CodeT, int D, int M>
static hipError_t tiled_launch_fmt(const GemmArgs &a, hipStream_t stream) {
    if constexpr (D == 8 && M == 1 && sizeof(CodeT) == 1) {
        const int split = choose_tiled_split(a);
        if (split > 0) {
            hipError_t e = tiled_launch_cfg<AT, CodeT, D, M, 2, 8, 4, 2>(a, stream, 0, split);          // 512 x 128
            if (e != hipSuccess) return e;
            return tiled_launch_cfg<AT, CodeT, D, M, 2, 4, 4, 2>(a, stream, split, 0x7fffffff);          // 256 x 128
        }
    }
    switch (choose_tiled_cfg(a)) {
    case 0:
        if constexpr (M == 2) return tiled_launch_cfg<AT, CodeT, D, M, 2, 4, 8, 1>(a, stream);   // 256 x 256, 8 token fragments a wave: every summed weight fragment feeds 8 MFMAs
        else if constexpr (D == 4 || D == 2) return tiled_launch_cfg<AT, CodeT, D, M, 2, 4, 8, 1>(a, stream);   // 256 x 256, 8 token fragments a wave
        else return tiled_launch_cfg<AT, CodeT, D, M, 4, 8, 4, 1>(a, stream);                    // 256 x 256
    case 1: return tiled_launch_cfg<AT, CodeT, D, M, 2, 4, 4, 2>(a, stream);   // 256 x 128
    case 2:
        if constexpr (M == 2 || D < 8) return tiled_launch_cfg<AT, CodeT, D, M, 2, 2, 8, 1>(a, stream);   // 128 x 256, 8 token fragments a wave
        else return tiled_launch_cfg<AT, CodeT, D, M, 2, 2, 4, 2>(a, stream);   // 128 x 128
    case 3: return tiled_launch_cfg<AT, CodeT, D, M, 1, 2, 4, 4>(a, stream);   // 256 x 64
    case 4: return tiled_launch_cfg<AT, CodeT, D, M, 1, 1, 4, 4>(a, stream);   // 128 x 64
    case 6:
        if constexpr (D == 8 && M == 1 && sizeof(CodeT) == 1) return tiled_launch_cfg<AT, CodeT, D, M, 2, 8, 4, 2>(a, stream);   // 512 x 128
        else return tiled_launch_cfg<AT, CodeT, D, M, 2, 4, 4, 2>(a, stream);
    default: return tiled_launch_cfg<AT, CodeT, D, M, 2, 1, 2, 4>(a, stream);  //  64 x 64
    }
}

template <typename AT>
static hipError_t tiled_launch_at(const GemmArgs &a, hipStream_t stream) {
    if (a.M == 2) return tiled_launch_fmt<AT, int8_t, 8, 2>(a, stream);
    if (a.code_bytes == 1) {
        switch (a.d) {
        case 8: return tiled_launch_fmt<AT, int8_t, 8, 1>(a, stream);
        case 4: return tiled_launch_fmt<AT, int8_t, 4, 1>(a, stream);
        default: return tiled_launch_fmt<AT, int8_t, 2, 1>(a, stream);
        }
    }
    switch (a.d) {
    case 8: return tiled_launch_fmt<AT, int16_t, 8, 1>(a, stream);
    case 4: return tiled_launch_fmt<AT, int16_t, 4, 1>(a, stream);
    default: return tiled_launch_fmt<AT, int16_t, 2, 1>(a, stream);
    }
}

}  // namespace pq
