// Fused codebook gather + MFMA GEMM for mid-size batches (about 32 to 512 tokens) on gfx950: d = 8, one codebook of at
// most 256 entries, one-byte codes in the PACKED order (pq_gemm_stream.hip), at most 8 chunks of 512 features.
//
// y[t][n] = scale[n] * sum_k x[t][k] * W[n][k] + bias[n],   W[n][8g..8g+7] = centroid(code[n][g])
//
// Between the decode kernel (16 tokens a block, every block gathers the weights again) and the tiled kernel (whole of K
// walked serially by one block per CU, which needs hundreds of large tiles to fill the chip) lie the batches a serving
// engine runs most: a few MFLOP per CU, where a launch is a few microseconds of fixed order around one microsecond
// of arithmetic and what a CU must TAKE IN -- its token tile's activations and its row tile's codes, from L2 at about
// 70 GB/s per CU -- decides the rest.  This kernel keeps the decode kernel's arithmetic (the weight fragment of
// v_mfma_f32_16x16x32 read straight out of the replicated LDS codebook image at the address the lane's code byte
// selects; the activations as B operands in registers, K split over the 8 waves of a block, partial sums summed in LDS
// in a fixed order) and changes the shape of a block's work:
//
//  * 32 tokens a block (two token fragments): the 32 operand registers a wave holds of one fragment in the decode
//    kernel become 2 x 64, every gathered weight fragment feeds two MFMAs (the LDS gather is then half of what the
//    matrix pipe can take instead of all of it), and a row tile's codes are streamed half as often;
//  * any number of rows a block: the wave's units (one 16-row fragment x its 512-feature chunk: 16 gathers + 32 MFMAs)
//    run in ROUNDS of nf fragments; after a round the waves along K sum that round's nf x 2 accumulator tiles through
//    LDS and store them, while the activations stay in registers and the code pieces keep streaming through a ring of
//    three KiB slots per wave (LDS-DMA, two units ahead, across the round boundaries).  So the activations of a
//    token tile are taken in once per block, whatever the block's height, and the height is chosen so that one round
//    of blocks fills the chip.
//
// Instruction fetch matters as in the decode kernel (a launch starts with a cold instruction cache): a unit is one loop
// body, rounds and fragments are run-time loops.
#pragma once
#include "pq_gemm_decode_impl.cuh"

namespace pq {

constexpr int WD_WAVES = 8;
constexpr int WD_WT = 2;                 // token fragments per block
constexpr int WD_TOKENS = 16 * WD_WT;    // tokens per block
constexpr int WD_MAX_NF = 4;             // fragments per wave and round
constexpr int WD_MAX_FRAGS = 64;         // fragments per block (its scales and bias sit in LDS)
constexpr int WD_WAVE_KIB = 11;          // LDS per wave behind the image, KiB: the code ring, then nf * WD_WT accumulator tiles
// code pieces a wave has in flight (1 KiB each): what the accumulator tiles leave, at least 3
constexpr int wd_ring(int nf) { return WD_WAVE_KIB - nf * WD_WT; }
constexpr size_t wd_wave_bytes(int nf) { return (size_t)(wd_ring(nf) + nf * WD_WT) * 1024; }
constexpr size_t wd_lds_bytes(size_t img_bytes, int nf, int frags_blk) {
    const size_t a = img_bytes + WD_WAVES * wd_wave_bytes(nf) + (size_t)frags_blk * 64;
    const size_t b = (size_t)WD_WAVES * 16 * DC_XROW_BYTES;           // the activation regions of the first phase
    return a > b ? a : b;
}

struct WideParams {
    const unsigned char *x;        // [T][x_stride] act dtype
    const unsigned char *codes;    // packed pieces
    const unsigned char *codebooks;
    const uint16_t *scales;
    const uint16_t *bias;          // nullable
    uint16_t *y;
    int64_t y_stride;              // elements
    uint32_t x_stride_bytes;
    int T;                         // tokens of the batch (32 per block)
    int nchunks;                   // 512-feature chunks of K (at most 1 << wk_log2)
    int K;                         // codebook entries (power of two, 16 .. 256)
    int nf;                        // fragments per wave and round (1 .. WD_MAX_NF)
    int rounds;                    // rounds per block
    int wk_log2;                   // log2 of the waves along K
    int ring;                      // code pieces a wave keeps in flight (LDS slots of 1 KiB)
    int n_tiles;                   // row tiles of the layer
    int tgroups;                   // token groups of 32
    uint32_t img_bytes;            // K * 256; the waves' areas follow
    int spt;                       // 16-byte image slots per thread
    uint32_t wave_bytes;           // LDS per wave: code ring, then nf * WD_WT accumulator tiles
    uint32_t sc_off;               // LDS offset of the tile's scales (uint16 per row), its bias follows
    TileMap tmap;                  // tiles of (8 >> wk_log2) * rounds * nf * 16 rows
};

__device__ __forceinline__ uint32_t wd_gload4(uint32_t voff, const void *sbase) {
    uint32_t v;
    asm volatile("; PQX %0\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}

template <int OFF>
__device__ __forceinline__ uint32_t wd_lds_read4_at(uint32_t addr) {
    uint32_t v;
    asm volatile("; PQX %0\n\tds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// STAMPS (diagnostic builds only): wave 0 and the last wave keep wall-clock stamps of the phases and store them at the
// end to `bias` (56 x u64 per block).  ABL (diagnostic builds only): bit 0 leaves the MFMAs out, bit 1 the gathers.
// NF: fragments per wave and round, a compile-time value so that the depth of the code ring -- and with it every counted
// wait of the unit loop -- is an immediate (a computed jump into a table of waits costs a unit about 80 cycles).
template <typename AT, int NF, bool STAMPS = false, int ABL = 0>
__global__ __launch_bounds__(WD_WAVES * 64) void gemm_wide_kernel(const WideParams p) {
    typedef typename Act<AT>::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint16_t *bias = p.bias;
    // Block b runs on XCD b % 8 (round-robin placement).  The token groups of one row tile stream the same codes: they
    // are consecutive blocks of ONE XCD, so that its L2 fetches a code piece from memory once for all of them.
    const int in_xcd = (int)blockIdx.x >> 3;
    const int row_tile = (in_xcd / p.tgroups) * 8 + ((int)blockIdx.x & 7);
    if (row_tile >= p.n_tiles) return;                    // the grid is rounded up to whole groups of 8 row tiles
    const int tok0 = (in_xcd % p.tgroups) * WD_TOKENS;
    const int T = p.T;
    unsigned long long st[12], sr[8], su[8];   // sr: inside the block's second round (the first of a one-round block); su: inside that round's second unit
    if constexpr (STAMPS) {
        bias = nullptr;
#pragma unroll
        for (int k = 0; k < 12; ++k) st[k] = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sr[k] = su[k] = 0;
        st[0] = __builtin_amdgcn_s_memrealtime();
    }
#define WD_USTAMP(k) if constexpr (STAMPS) { if (r == (rounds > 1 ? 1 : 0) && i == (nf > 1 ? 1 : 0)) { asm volatile("" ::: "memory"); su[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } }
#define WD_RSTAMP(k) if constexpr (STAMPS) { if (r == (rounds > 1 ? 1 : 0)) { asm volatile("" ::: "memory"); sr[k] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); } }
#define WD_STAMP(k) if constexpr (STAMPS) { asm volatile("" ::: "memory"); st[k] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); }

    const int WK = 1 << p.wk_log2, WR = WD_WAVES >> p.wk_log2;
    const int wk = wave & (WK - 1), wr = wave >> p.wk_log2;
    constexpr int nf = NF;
    const int rounds = p.rounds;
    const bool has_chunk = wk < p.nchunks;                // a wave beyond the last chunk contributes zeros

    // ---- the block's tile
    const int frags_blk = WR * rounds * nf;
    const TileLoc loc = locate_tile(p.tmap, row_tile, frags_blk * 16);
    const int nfr = loc.nrows >> 4;                       // valid fragments of the tile (partitions are multiples of 16 rows)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    WD_STAMP(1)   // tile found

    // ---- codebook entries (as in the decode kernel): entry e fills 16 image slots of 16 bytes, a thread owns spt
    // consecutive slots.  Requested first, stored once the activations are out of the LDS.
    const int spt = p.spt;
    const bool img_active = tid * spt < (p.K << 4);
    const int ncb = spt > 16 ? spt >> 4 : 1;
    const int e0 = img_active ? (tid * spt) >> 4 : 0;
    const unsigned char *cb_part = p.codebooks + (size_t)loc.part * p.K * 16;
    u32x4 cbv0 = dc_gload16<0>((uint32_t)e0 * 16u, cb_part), cbv1, cbv2, cbv3;
    if (ncb > 1) {
        cbv1 = dc_gload16<16>((uint32_t)e0 * 16u, cb_part);
        cbv2 = dc_gload16<0>((uint32_t)(e0 + (ncb > 2 ? 2 : 0)) * 16u, cb_part);
        cbv3 = dc_gload16<16>((uint32_t)(e0 + (ncb > 2 ? 2 : 0)) * 16u, cb_part);
    }
    // scales and bias of rows 2 tid, 2 tid + 1 of the tile (rows past the tile's last repeat its last pair)
    const int rpair = 2 * tid < loc.nrows ? 2 * tid : loc.nrows - 2;
    uint32_t scw = wd_gload4((uint32_t)(loc.row0 + rpair) * 2u, p.scales);
    uint32_t biw = wd_gload4((uint32_t)(loc.row0 + rpair) * 2u, bias ? bias : p.scales);   // without a bias: loaded, never used

    // ---- activations: the wave's chunk of 16 tokens at a time, one contiguous KiB per token by LDS-DMA into the wave's
    // padded region (see the decode kernel), read from there into the 16 operand registers of the token fragment.
    // Tokens past T repeat the last one (their outputs are never stored).  A block whose second token fragment is empty
    // uses the first one's registers for it.
    const uint32_t xreg = lds0 + (uint32_t)wave * (16u * DC_XROW_BYTES);
    const unsigned char *xchunk = p.x + (size_t)(has_chunk ? wk : 0) * 1024;
    const uint32_t xrd = xreg + (uint32_t)(lane & 15) * DC_XROW_BYTES + (uint32_t)(lane >> 4) * 16u;
    u32x4 B0[16], B1[16];
    auto fetch_tokens = [&](int first) {
        uint32_t xdst = xreg;
        const unsigned char *src = xchunk + (size_t)first * p.x_stride_bytes;
        int left = T - first - 1;                         // rows after this one that exist
#pragma nounroll
        for (int t = 0; t < 16; ++t) {
            dc_dma16(xdst, lane16, src);
            xdst += DC_XROW_BYTES;
            if (left > 0) src += p.x_stride_bytes;
            --left;
        }
    };
    fetch_tokens(tok0);
    WD_STAMP(2)   // entries and first activations requested
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dc_landed(cbv0);
    if (ncb > 1) { dc_landed(cbv1); dc_landed(cbv2); dc_landed(cbv3); }
    dc_landed(scw);
    dc_landed(biw);
#define WD_XR(Bv, s) Bv[s] = dc_lds_read16_at<64 * (s)>(xrd);
#define WD_XD(Bv, s) dc_landed(Bv[s]);
#define WD_ALL16(M, Bv) M(Bv, 0) M(Bv, 1) M(Bv, 2) M(Bv, 3) M(Bv, 4) M(Bv, 5) M(Bv, 6) M(Bv, 7) M(Bv, 8) M(Bv, 9) M(Bv, 10) M(Bv, 11) M(Bv, 12) M(Bv, 13) M(Bv, 14) M(Bv, 15)
    WD_ALL16(WD_XR, B0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WD_ALL16(WD_XD, B0)
    WD_STAMP(3)   // first token fragment in registers
    if (tok0 + 16 < T) {
        fetch_tokens(tok0 + 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WD_ALL16(WD_XR, B1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WD_ALL16(WD_XD, B1)
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) B1[s] = B0[s];
    }
#undef WD_XR
#undef WD_XD
#undef WD_ALL16
    WD_STAMP(4)   // operands in registers
    __syncthreads();   // every wave has its activations out of the LDS: image, rings and accumulator tiles may be written
    WD_STAMP(5)

    // ---- the wave's units: unit u is chunk wk of fragment (r * WR + wr) * nf + i of the tile, r = u / nf, i = u % nf.
    // Its piece comes by DMA into ring slot u % 3, two units ahead of its use; units past the last repeat it (same
    // counts, no branch around a load), fragments past the tile's last repeat that one and are never stored.
    const int UT = rounds * nf;
    const uint32_t ring0 = lds0 + p.img_bytes + (uint32_t)wave * p.wave_bytes;
    constexpr int ahead = wd_ring(NF);                    // a unit's piece is requested `ahead` units before its use
    const uint32_t ring_last = ring0 + (uint32_t)(ahead - 1) * 1024u;
    const size_t frag_bytes = (size_t)p.nchunks * 1024;
    const unsigned char *code_base = p.codes + ((size_t)(loc.row0 >> 4) * p.nchunks + (has_chunk ? wk : 0)) * 1024;
    // the next piece to request: unit `du`, whose fragment is `dfr` of the tile, at byte offset `doff` from code_base (a
    // tile's pieces span at most 64 fragments x 8 KiB); requests past the last unit repeat it, fragments past the tile's
    // last repeat that one
    int du = 0, di = 0;
    uint32_t doff = (uint32_t)(wr * nf) * (uint32_t)frag_bytes;
    const uint32_t doff_max = (uint32_t)(nfr - 1) * (uint32_t)frag_bytes;
    const uint32_t dstep_round = (uint32_t)((WR - 1) * nf + 1) * (uint32_t)frag_bytes;   // from a round's last fragment to the next round's first
    uint32_t dslot = ring0;
    auto dma_unit = [&]() {
        dc_dma16(dslot, lane16, code_base + (doff < doff_max ? doff : doff_max));
        if (du + 1 < UT) {
            ++du;
            if (++di == nf) { di = 0; doff += dstep_round; } else doff += (uint32_t)frag_bytes;
        }
        dslot = dslot == ring_last ? ring0 : dslot + 1024u;
    };
    if (has_chunk) {
#pragma nounroll
        for (int k = 0; k < ahead; ++k) dma_unit();
    }
    // scales and bias of the tile's rows -> LDS, for the epilogue of every round
    if (2 * tid < frags_blk * 16) {
        *reinterpret_cast<uint32_t *>(smem + p.sc_off + (size_t)tid * 4) = scw;
        *reinterpret_cast<uint32_t *>(smem + p.sc_off + (size_t)frags_blk * 32 + (size_t)tid * 4) = biw;
    }
    // ---- codebook image
    if (img_active) {
        if (spt == 8) {
            const uint32_t a0 = (uint32_t)tid * 128u + ((uint32_t)(tid & 7) << 4);
#define WD_IS(c) *reinterpret_cast<u32x4 *>(smem + (a0 ^ ((c) << 4))) = cbv0;
            WD_IS(0) WD_IS(1) WD_IS(2) WD_IS(3) WD_IS(4) WD_IS(5) WD_IS(6) WD_IS(7)
#undef WD_IS
        } else {
            const int spe = spt > 16 ? 16 : spt;
            auto put = [&](const u32x4 &v, int base) {
#pragma nounroll
                for (int c = 0; c < spe; ++c)
                    *reinterpret_cast<u32x4 *>(smem + (size_t)(base + ((c + tid) & (spe - 1))) * 16) = v;
            };
            const int first = tid * spt;
            put(cbv0, first);
            if (ncb > 1) {
                put(cbv1, first + 16);
                if (ncb > 2) {
                    put(cbv2, first + 32);
                    put(cbv3, first + 48);
                }
            }
        }
    }
    const uint32_t ring_bytes = (uint32_t)ahead * 1024u;
    unsigned char *acc_tiles = smem + p.img_bytes + (size_t)wave * p.wave_bytes + ring_bytes;   // [nf][WD_WT] KiB
    if (!has_chunk) {
#pragma nounroll
        for (int i = 0; i < nf * WD_WT; ++i)
            *reinterpret_cast<f32x4 *>(acc_tiles + ((size_t)i * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // the image is complete
    WD_STAMP(6)   // image stored, barrier

    const uint32_t img_lane = lds0 + (uint32_t)(lane & 15) * 16u;   // this lane's copy of entry 0
    const uint32_t sel0 = 0x0c0c0400u, sel1 = 0x0c0c0500u, sel2 = 0x0c0c0600u, sel3 = 0x0c0c0700u;
    // Units are software-pipelined over two sets of 8 gathered weight fragments: the gathers of one half of a unit's
    // k-steps -- or of the next unit's first half -- go out one per step between the MFMA pairs of the other half, so
    // that a wave's instruction stream is MFMAs throughout (two waves share a SIMD's matrix pipe: whenever one of them
    // is in a stretch without MFMAs the pipe idles unless the other happens to be in one with).  LDS reads return in
    // order; every count is the number of LDS operations issued behind the one waited for (or fewer: waiting for more
    // is safe).
    u32x4 a0[8], a1[8];
    uint32_t cw[4], cn[4];   // the lane's 16 code bytes of the current unit's piece, and of the next one's (in flight)
    uint32_t cslot = ring0;
#define WD_G(A, j, s, sel) if constexpr (ABL & 2) { A[j] = u32x4{cw[0], cw[1], cw[2], cw[3]}; asm volatile("; PQX %0" : "+v"(A[j])); } else A[j] = dc_lds_read16(__builtin_amdgcn_perm(cw[(s) >> 2], img_lane, sel));
// The next piece's codes come as four dword reads into registers of their own, and move into `cw` by instructions
// written here, behind the wait: a copy the compiler makes of a register whose read is still in flight can land in
// front of the wait (it knows nothing of the read), and a loop-carried value always needs one.
#define WD_READ_CODES(slot) { const uint32_t ca_ = (slot) + lane16; cn[0] = wd_lds_read4_at<0>(ca_); cn[1] = wd_lds_read4_at<4>(ca_); cn[2] = wd_lds_read4_at<8>(ca_); cn[3] = wd_lds_read4_at<12>(ca_); }
#define WD_TAKE_CODES(n)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(%8)\n\t; PQL %4\n\t; PQL %5\n\t; PQL %6\n\t; PQL %7\n\t"                      \
                 "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"                    \
                 : "=&v"(cw[0]), "=&v"(cw[1]), "=&v"(cw[2]), "=&v"(cw[3])                                              \
                 : "v"(cn[0]), "v"(cn[1]), "v"(cn[2]), "v"(cn[3]), "n"(n) : "memory");
#define WD_GLO WD_G(a0, 0, 0, sel0) WD_G(a0, 1, 1, sel1) WD_G(a0, 2, 2, sel2) WD_G(a0, 3, 3, sel3) WD_G(a0, 4, 4, sel0) WD_G(a0, 5, 5, sel1) WD_G(a0, 6, 6, sel2) WD_G(a0, 7, 7, sel3)
#define WD_GHI WD_G(a1, 0, 8, sel0) WD_G(a1, 1, 9, sel1) WD_G(a1, 2, 10, sel2) WD_G(a1, 3, 11, sel3) WD_G(a1, 4, 12, sel0) WD_G(a1, 5, 13, sel1) WD_G(a1, 6, 14, sel2) WD_G(a1, 7, 15, sel3)
// (two accumulators per token fragment, even and odd k-steps: an MFMA that adds to the result of the one before it
// waits for that result -- four pipe passes -- so two chains a wave keep the pipe half busy, four keep it fed)
#define WD_MM(A, j, s)                                                                                              \
    if constexpr (ABL & 1) { accA[(s) & 1].x += __uint_as_float(A[j].x ^ B0[s].x); accB[(s) & 1].x += __uint_as_float(A[j].y ^ B1[s].x); } else { \
    accA[(s) & 1] = Act<AT>::mfma16(__builtin_bit_cast(vec8, A[j]), __builtin_bit_cast(vec8, B0[s]), accA[(s) & 1]);  \
    accB[(s) & 1] = Act<AT>::mfma16(__builtin_bit_cast(vec8, A[j]), __builtin_bit_cast(vec8, B1[s]), accB[(s) & 1]); } \
    __builtin_amdgcn_sched_barrier(0);
#define WD_MW(j, s, n) asm volatile("s_waitcnt lgkmcnt(%1)\n\t; PQL %0" : "+v"(a0[j]) : "n"(n)); WD_MM(a0, j, s)
#define WD_MN(j, s) WD_MM(a1, j, s)
    // (defined on every path, so that no path gets a compiler-made initialisation behind the first hand-issued reads)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "=v"(a0[j]), "=v"(a1[j]));
    asm volatile("" : "=v"(cw[0]), "=v"(cw[1]), "=v"(cw[2]), "=v"(cw[3]), "=v"(cn[0]), "=v"(cn[1]), "=v"(cn[2]), "=v"(cn[3]));
    if (has_chunk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ahead - 1) : "memory");   // unit 0's piece (the later ones may still be on their way)
        WD_READ_CODES(cslot)
        WD_TAKE_CODES(0)
        WD_GLO
    }
    WD_STAMP(7)   // first gathers issued
#pragma nounroll
    for (int r = 0; r < rounds; ++r) {
        WD_RSTAMP(0)   // round starts
        if (has_chunk) {
#pragma nounroll
            for (int i = 0; i < nf; ++i) {
                if (i == 1) { WD_RSTAMP(1) }   // first unit done
                WD_USTAMP(0)
                dma_unit();             // unit u + ahead into the slot this unit's piece has left
                f32x4 accA[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accB[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                // The next unit's piece: behind it the ahead - 1 younger requests.  The counter also counts the stores
                // of a round's epilogue, in order with the loads: so that the first unit of a round does not wait for
                // stores issued a moment ago, the last unit of the round before it has waited for one piece more (this
                // wait then covers the piece the first unit reads), and the first unit waits for nothing.
                if (i == nf - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ahead - 2) : "memory");
                else if (!(i == 0 && r > 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ahead - 1) : "memory");
                const uint32_t nslot = cslot == ring_last ? ring0 : cslot + 1024u;
                WD_READ_CODES(nslot)
                WD_USTAMP(1)
                // First half of the k-steps: per step the gather of the matching step of the second half goes out, then
                // the two MFMAs of the step.  Behind a0[j] at its wait: 7 - j of its set, (the two tile stores of the
                // unit before,) the four reads above and j + 1 of the other set -- at least 12.
#define WD_S1(j, s, sel) WD_G(a1, j, (s) + 8, sel) WD_MW(j, s, 12)
                WD_S1(0, 0, sel0) WD_S1(1, 1, sel1) WD_S1(2, 2, sel2) WD_S1(3, 3, sel3)
                WD_S1(4, 4, sel0) WD_S1(5, 5, sel1) WD_S1(6, 6, sel2) WD_S1(7, 7, sel3)
#undef WD_S1
                WD_USTAMP(4)
                WD_TAKE_CODES(8)         // the next unit's codes: behind them the 8 gathers above
                cslot = nslot;
                WD_USTAMP(5)
                // Second half: per step the gather of the NEXT unit's first half (of this unit again after the last:
                // never used), then the step's MFMAs.  Behind a1[j]: 7 - j of its set and j + 1 of the other: 8.
#define WD_S2(j, s, sel) WD_G(a0, j, j, sel) asm volatile("s_waitcnt lgkmcnt(8)\n\t; PQL %0" : "+v"(a1[j])); WD_MM(a1, j, s)
                WD_S2(0, 8, sel0) WD_S2(1, 9, sel1) WD_S2(2, 10, sel2) WD_S2(3, 11, sel3)
                WD_S2(4, 12, sel0) WD_S2(5, 13, sel1) WD_S2(6, 14, sel2) WD_S2(7, 15, sel3)
#undef WD_S2
                *reinterpret_cast<f32x4 *>(acc_tiles + ((size_t)(i * WD_WT) * 64 + lane) * 16) = accA[0] + accA[1];
                *reinterpret_cast<f32x4 *>(acc_tiles + ((size_t)(i * WD_WT + 1) * 64 + lane) * 16) = accB[0] + accB[1];
                WD_USTAMP(7)
            }
            // the gathers issued behind the round's last unit: landed before any branch
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define WD_L0(j) dc_landed(a0[j]);
            WD_L0(0) WD_L0(1) WD_L0(2) WD_L0(3) WD_L0(4) WD_L0(5) WD_L0(6) WD_L0(7)
#undef WD_L0
        }
        if (r == 0) { WD_STAMP(8) }   // first round's units done
        WD_RSTAMP(2)   // units done
        __syncthreads();               // every wave's tiles of this round are in LDS
        WD_RSTAMP(3)   // barrier passed

        // ---- the round's nf x 2 output tiles of row group wr: wave wk sums tiles wk, wk + WK, ... over the waves along
        // K in a fixed order; scale, bias, one rounding, 8 bytes stored per lane (token lane & 15, rows 4 (lane >> 4) .. + 3).
        // With 4 or 8 waves along K a wave has at most two tiles: it sums them between the two barriers and finishes and
        // stores them behind the second, when the other waves are in the next round's units already.
        auto tile_sum = [&](int tt) {
            const unsigned char *slot = smem + p.img_bytes + (size_t)(wr << p.wk_log2) * p.wave_bytes + ring_bytes +
                                        ((size_t)tt * 64 + lane) * 16;
            f32x4 sum = *reinterpret_cast<const f32x4 *>(slot);
            if (WK >= 4) {   // four reads in flight at a time (each is an LDS round trip), added in wave order
#pragma nounroll
                for (int w = 0; w < WK; w += 4) {
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(slot + (size_t)(w + 1) * p.wave_bytes);
                    const f32x4 v2 = *reinterpret_cast<const f32x4 *>(slot + (size_t)(w + 2) * p.wave_bytes);
                    const f32x4 v3 = *reinterpret_cast<const f32x4 *>(slot + (size_t)(w + 3) * p.wave_bytes);
                    if (w) {
                        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(slot + (size_t)w * p.wave_bytes);
                        sum.x += v0.x; sum.y += v0.y; sum.z += v0.z; sum.w += v0.w;
                    }
                    sum.x += v1.x; sum.y += v1.y; sum.z += v1.z; sum.w += v1.w;
                    sum.x += v2.x; sum.y += v2.y; sum.z += v2.z; sum.w += v2.w;
                    sum.x += v3.x; sum.y += v3.y; sum.z += v3.z; sum.w += v3.w;
                }
            } else {
#pragma nounroll
                for (int w = 1; w < WK; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(slot + (size_t)w * p.wave_bytes);
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                }
            }
            return sum;
        };
        auto tile_store = [&](int tt, const f32x4 &sum) {
            const int i = tt >> 1, j = tt & 1;
            const int fr = (r * WR + wr) * nf + i;
            const int tok = tok0 + 16 * j + (lane & 15);
            if (fr < nfr && tok < T) {
                const int rl = fr * 16 + (lane >> 4) * 4;       // first of the lane's four rows, within the tile
                const u32x2 scv = *reinterpret_cast<const u32x2 *>(smem + p.sc_off + (size_t)rl * 2);
                const u32x2 biv = *reinterpret_cast<const u32x2 *>(smem + p.sc_off + (size_t)frags_blk * 32 + (size_t)rl * 2);
                const uint32_t sw[2] = {scv.x, scv.y}, bw[2] = {biv.x, biv.y};
                uint16_t h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t s16 = (uint16_t)(sw[e >> 1] >> ((e & 1) * 16)), b16 = (uint16_t)(bw[e >> 1] >> ((e & 1) * 16));
                    h[e] = Act<AT>::from_f32(sum[e] * Act<AT>::to_f32(s16) + (bias ? Act<AT>::to_f32(b16) : 0.0f));
                }
                u32x2 o2;
                o2.x = h[0] | ((uint32_t)h[1] << 16);
                o2.y = h[2] | ((uint32_t)h[3] << 16);
                *reinterpret_cast<u32x2 *>(p.y + (size_t)tok * p.y_stride + loc.row0 + rl) = o2;
            }
        };
        const int ntile = nf * WD_WT;
        if (WK >= 4) {
            const bool h0 = wk < ntile, h1 = wk + WK < ntile;   // wave-uniform
            f32x4 s0{0.f, 0.f, 0.f, 0.f}, s1{0.f, 0.f, 0.f, 0.f};
            if (h0) s0 = tile_sum(wk);
            if (h1) s1 = tile_sum(wk + WK);
            if constexpr (STAMPS) asm volatile("s_nop 0" :: "v"(s0.x), "v"(s1.x));
            WD_RSTAMP(4)   // tiles summed
            if (r + 1 < rounds) __syncthreads();   // the tiles have been read: the next round may overwrite them
            WD_RSTAMP(5)   // second barrier passed
            if (h0) tile_store(wk, s0);
            if (h1) tile_store(wk + WK, s1);
            WD_RSTAMP(6)   // stores issued
        } else {
#pragma nounroll
            for (int tt = wk; tt < ntile; tt += WK) tile_store(tt, tile_sum(tt));
            if (r + 1 < rounds) __syncthreads();
        }
    }
#undef WD_G
#undef WD_READ_CODES
#undef WD_TAKE_CODES
#undef WD_GLO
#undef WD_GHI
#undef WD_MM
#undef WD_MW
#undef WD_MN
    WD_STAMP(9)   // all rounds done
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[11] = __builtin_amdgcn_s_memrealtime();   // stores retired
        if (lane == 0 && (wave == 0 || wave == WD_WAVES - 1)) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<uint16_t *>(p.bias)) +
                                    (size_t)blockIdx.x * 56 + (wave == 0 ? 0 : 28);
#pragma unroll
            for (int k = 0; k < 12; ++k) o[k] = st[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[12 + k] = sr[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[20 + k] = su[k];
        }
    }
#undef WD_STAMP
#undef WD_RSTAMP
#undef WD_USTAMP
}

}  // namespace pq
