// Fused codebook gather + MFMA GEMM for small and medium token counts (decode batches) on gfx950.
//
// y[t][n] = scale[n] * sum_k x[t][k] * W[n][k] + bias[n],   W[n][8g..8g+7] = centroid(code[n][g])
//
// Shape of the computation.  The reconstructed weight is the A operand of
// v_mfma_f32_16x16x32_{bf16,f16}: lane l supplies A[row l&15][k = 8(l>>4) .. +7], i.e. eight
// consecutive k of ONE weight row -- exactly one d=8 centroid (two d=4, four d=2).  So a lane's
// A fragment is read straight out of the LDS codebook with one ds_read_b128 at the address its code
// byte selects: no shuffle, no staging of W.  The activations are the B operand (lane l supplies
// x[t = l&15][the same 8 k]) and stay in registers: they are loaded once per k-chunk and reused by
// every weight fragment of the tile.  A 16x16 accumulator holds y^T for 16 rows x 16 tokens.
//
// Work split.  A block owns NF weight fragments (16*NF rows of one output partition) x TF token
// fragments (16*TF tokens) and all of K; its 8 waves split K in chunks (one chunk = what a lane's
// 16-byte code load covers, times 4 lane quads), each wave streaming its codes from HBM with 16-byte
// loads and its slice of the activations from L2 straight into registers (32 registers at a time,
// one sub-chunk ahead).  There is no barrier in the main loop.  A gathered weight fragment feeds TF
// MFMAs, an activation fragment NF.  The partial accumulators of the waves are summed in a fixed order
// through LDS (deterministic), then scaled, biased and stored; scales and bias are fetched at kernel
// start so that no global load sits on the tail.
//
// Code layouts.  NATURAL is the reference's [N][G][M]: a lane's 16 bytes are 16 consecutive groups of
// its row, so MFMA k-step s of lane quad q covers groups 16q+s and the matching activation loads are
// 16-byte pieces 128 B apart (poorly coalesced).  PACKED is this library's load-time re-layout
// (pq_repack_codes): one 1-KiB piece per (16-row fragment, chunk), lane-linear, with byte s of lane
// (r, q) = group 4s+q of row r.  k-step s then covers 32 consecutive k, the activation loads are
// contiguous 64-byte runs per token, and a wave's code load is one contiguous KiB.
//
// LDS codebook image.  Entry e is stored R times back to back; lane l reads copy l mod R.  With
// R = 16 (d=8), 32 (d=4, d=2) every lane of an LDS lane group reads a different bank slot whatever
// the codes are, so the gather runs at the conflict-free ds_read rate.
#pragma once
#include "pq_common.cuh"
#include "pq_kernels.h"
#include "pq_gather.cuh"
#include "pq_stream_geom.cuh"
#include <atomic>

namespace pq {

constexpr int ST_THREADS = 512;
constexpr int ST_WAVES = ST_THREADS / 64;
// accumulator fragments summed across the waves per round: the largest divisor of the tile's count up to 8
constexpr int stream_reduce_frags(int fr) { return fr <= 8 ? fr : fr % 8 == 0 ? 8 : fr % 7 == 0 ? 7 : fr % 6 == 0 ? 6 : 4; }

// ABL (diagnostic builds only; 0 in the library): bit 0 skips the LDS gather, bit 1 the MFMAs, bit 2 the
// code loads, bit 3 the activation loads, bit 4 the codebook image, bit 5 the cross-wave sum + stores.
// ACT (fused SiLU-and-mul over a gate/up pair of equal partitions, act_rows rows each): the block's first NF/2
// fragments are rows of the gate partition, the other NF/2 the same rows of the up partition, both codebooks
// sit in the LDS image, and the epilogue stores y[t][n] = silu(gate[t][n]) * up[t][n], n < act_rows -- each
// factor rounded to the act dtype first, exactly as the unfused GEMM followed by silu_and_mul rounds them.
// COALT: the activations are loaded with the lanes along the rows and placed through LDS (packed layout, from 16 tokens on)
template <typename AT, typename CodeT, int D, int M, int NF, int TF, bool PACKED, int ABL = 0, bool ACT = false, bool COALT = false>
__global__ __launch_bounds__(ST_THREADS) void gemm_stream_kernel(
    const uint16_t *__restrict__ x, int T, int64_t x_stride, const void *__restrict__ codes_v,
    const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales, const uint16_t *__restrict__ bias,
    TileMap tmap, int G, int K, int nbits, int R, uint16_t *__restrict__ y, int64_t y_stride, int ksplit,
    float *__restrict__ ws_part, unsigned *__restrict__ ws_cnt, int act_rows = 0) {
    typedef typename Act<AT>::vec8 vec8;
    typedef StreamFmt<sizeof(CodeT), D, M> F;
    constexpr int CB = sizeof(CodeT);
    constexpr int EB = 2 * D;
    constexpr int GPL = F::GPL, CG = F::CG, KS = F::KS, CPS = F::CPS;
    constexpr int SUBK = (8 / TF) < KS ? (8 / TF) : KS;   // k-steps per activation sub-load: <= 32 registers
    constexpr int NSUB = KS / SUBK;
    constexpr int FR = NF * TF;                            // accumulator fragments per wave
    constexpr int RF = stream_reduce_frags(FR);            // fragments per reduction round (<= 64 KiB of LDS)
    constexpr int ROUNDS = FR / RF;
    static_assert(KS % SUBK == 0 && FR % RF == 0, "unsupported tile shape");
    constexpr int NFH = ACT ? NF / 2 : NF;                 // fragments of distinct output rows per block
    static_assert(!ACT || (TF == 1 && NF % 2 == 0 && NF <= 8 && M == 1 && PACKED), "fused activation: one round, M = 1");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ABL bit 6 (diagnostic build): thread 0 of every block writes wall-clock stamps of the phases into
    // the buffer passed as `bias` (8 x u64 per block); such a build computes no usable output.
    unsigned long long *stamps = nullptr;
    if constexpr (ABL & 64) {
        stamps = reinterpret_cast<unsigned long long *>(const_cast<uint16_t *>(bias)) +
                 (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
        bias = nullptr;
        if (tid == 0) stamps[0] = wall_clock64();
    }
    const int r16 = lane & 15;   // weight row within a fragment (A) / token within a fragment (B)
    const int q = lane >> 4;     // k quad

    // K may be split over `ksplit` blocks per tile (grid.x = ksplit * tiles): block z of a tile takes a
    // contiguous range of chunks and the tile's last block to finish adds the partial sums up (see the end)
    const int n_tiles_x = (int)gridDim.x / ksplit;
    const int zsplit = (int)blockIdx.x / n_tiles_x;
    const int tile_x = (int)blockIdx.x - zsplit * n_tiles_x;
    const TileLoc loc = locate_tile(tmap, tile_x, NFH * 16);   // ACT: the map covers the gate partition only
    const int t0 = blockIdx.y * (16 * TF);
    const int row_end = loc.row0 + loc.nrows;
    const int Kin = G * D;
    const int nchunks = (G + CG - 1) / CG;
    const int c_begin = zsplit * nchunks / ksplit, c_end = (zsplit + 1) * nchunks / ksplit;   // this block's chunks

    f32x4 acc[NF][TF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < TF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Tokens past T read the last valid token's row (same addresses, so no extra traffic) and are
    // zeroed with a select: no divergent branch around any load.
    bool tok_ok[TF];
    const uint16_t *xrow[TF];
#pragma unroll
    for (int j = 0; j < TF; ++j) {
        const int t = t0 + 16 * j + r16;
        tok_ok[j] = t < T;
        xrow[j] = x + (size_t)(t < T ? t : T - 1) * x_stride;
    }

    // Packed layout: the activations are LOADED with the lanes along the rows -- lane l takes 16-byte piece (l & 3) of
    // token (l >> 2) of a k-step, so every quad of lanes reads 64 contiguous bytes; loaded in the operand's own shape
    // (lane (q, c) = token c, k-octet q: the four lanes of a quad on four rows) the memory pipeline takes a wave-load
    // at well under half the rate (tools/ingest_lab.hip) -- and turned into the MFMA operand through a KiB of LDS per
    // register of the wave's own (place_x): written at slot u(token, piece), read back at u(c, q).  u is chosen so that
    // neither the 8-lane groups of the write nor the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) meet
    // on a bank: u = c0 + 2 q0 + 4 (q1 ^ c1) + 8 c3 + 16 (c2 + 2 c1).
    // Taken from 16 tokens on (COALT, the launcher's choice): below, the round trip through LDS costs a launch of few
    // tokens more than the loads save (measured: 4 - 9 % slower at 1 - 8 tokens, 3 - 11 % faster at 16 - 256).
    constexpr bool COAL = COALT && PACKED && !ACT && !(ABL & 8);
    auto x_slot = [](int c, int qq) {
        return (uint32_t)((c & 1) + 2 * (qq & 1) + 4 * (((qq >> 1) ^ (c >> 1)) & 1) + 8 * ((c >> 3) & 1) +
                          16 * (((c >> 2) & 1) + 2 * ((c >> 1) & 1)));
    };
    const int lt = lane >> 2, lp = lane & 3;
    bool ltok_ok[TF];
    const uint16_t *lrow[TF];
#pragma unroll
    for (int j = 0; j < TF; ++j) {
        const int t = t0 + 16 * j + lt;
        ltok_ok[j] = t < T;
        lrow[j] = x + (size_t)(t < T ? t : T - 1) * x_stride;
    }
    // the wave's KiB slots lie behind the codebook image and the reduction area
    const uint32_t x_lds_off = COAL ? stream_x_lds_offset((size_t)M * K * EB * R, (size_t)ST_WAVES * RF * 64 * 16) : 0u;
    unsigned char *x_lds = smem + x_lds_off + (size_t)wave * (SUBK * TF * 1024);
    const uint32_t x_wr = x_slot(lt, lp) * 16u, x_rd = x_slot(r16, q) * 16u;

    struct Codes { u32x4 cw[NF]; };
    struct XSub { vec8 f[SUBK][TF]; };

    auto load_codes = [&](Codes &cd, int c) {
        if constexpr (PACKED) {
            // [fragment][chunk][lane] 16-byte pieces; fragments are 16 absolute rows
            const u32x4 *pk = static_cast<const u32x4 *>(codes_v);
            const int f0 = loc.row0 >> 4;
            const int nfr = (loc.nrows + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int ih = ACT ? i % NFH : i;             // ACT: fragments NFH.. are the up rows, act_rows further on
                const int f = f0 + (ih < nfr ? ih : nfr - 1) + (ACT && i >= NFH ? act_rows >> 4 : 0);   // clamp: rows past the tile are never stored
                if constexpr (ABL & 4) cd.cw[i] = u32x4{(uint32_t)(f * 977 + lane), (uint32_t)c, (uint32_t)lane * 31, 7u};
                else cd.cw[i] = __builtin_nontemporal_load(&pk[((size_t)f * nchunks + c) * 64 + lane]);
            }
        } else {
            const CodeT *codes = static_cast<const CodeT *>(codes_v);
            const int g0 = c * CG + GPL * q;           // first input group of this lane
            const int g0c = g0 < G ? g0 : 0;           // G % GPL == 0: all-or-nothing per lane
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                int n = loc.row0 + 16 * i + r16;
                n = n < row_end ? n : row_end - 1;
                cd.cw[i] = __builtin_nontemporal_load(
                    reinterpret_cast<const u32x4 *>(codes + ((size_t)n * G + g0c) * M));
            }
        }
    };
    // activation fragments of k-steps sub*SUBK .. +SUBK-1 of chunk c (COAL: as loaded, place_x makes operands of them)
    auto load_x = [&](XSub &xs, int c, int sub) {
#pragma unroll
        for (int s = 0; s < SUBK; ++s) {
            int k;
            if constexpr (PACKED) k = c * (CG * D) + 32 * (sub * SUBK + s) + 8 * (COAL ? lp : q);   // 32 consecutive k per step (compile-time choice)
            else k = (c * CG + GPL * q) * D + 8 * (sub * SUBK + s);                 // the lane's own 16 groups
            const bool k_ok = k < Kin;
#pragma unroll
            for (int j = 0; j < TF; ++j) {
                u32x4 v;
                if constexpr (ABL & 8) v = u32x4{(uint32_t)k, 1u, 2u, (uint32_t)lane};
                else v = *reinterpret_cast<const u32x4 *>((COAL ? lrow[j] : xrow[j]) + (k_ok ? k : 0));
                // zeroing by mask, not by select: a select lets the compiler sink the load into a branch
                const uint32_t keep = 0u - (uint32_t)(k_ok && (COAL ? ltok_ok[j] : tok_ok[j]));
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                xs.f[s][j] = __builtin_bit_cast(vec8, v);
            }
        }
    };
    // COAL: loaded registers -> MFMA operands, through the wave's LDS slots (a wave's LDS accesses execute in order)
    auto place_x = [&](XSub &xs) {
        if constexpr (COAL) {
#pragma unroll
            for (int s = 0; s < SUBK; ++s)
#pragma unroll
                for (int j = 0; j < TF; ++j)
                    *reinterpret_cast<u32x4 *>(x_lds + (s * TF + j) * 1024 + x_wr) = __builtin_bit_cast(u32x4, xs.f[s][j]);
#pragma unroll
            for (int s = 0; s < SUBK; ++s)
#pragma unroll
                for (int j = 0; j < TF; ++j)
                    xs.f[s][j] = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4 *>(x_lds + (s * TF + j) * 1024 + x_rd));
        }
    };

    const uint32_t shift = (uint32_t)__builtin_ctz((unsigned)(EB * R));   // log2 of the entry stride
    const uint32_t lane_off = (uint32_t)(lane & (R - 1)) * EB;
    const uint32_t cb_stride = (uint32_t)K * EB * R;                       // bytes per codebook image
    auto compute_sub = [&](const Codes &cd, const XSub &xs, int sub) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int s = 0; s < SUBK; ++s) {
                const int ss = sub * SUBK + s;   // k-step within the chunk
                vec8 a;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    uint32_t idx[CPS];
#pragma unroll
                    for (int u = 0; u < CPS; ++u)
                        idx[u] = code_at<CB>(cd.cw[i], (ss * CPS + u) * M + m, (uint32_t)nbits);
                    vec8 am;
                    if constexpr (ABL & 1) {
                        u32x4 fake{idx[0], 0u, 0u, 0u};
                        am = __builtin_bit_cast(vec8, fake);
                    } else {
                        am = gather_frag<AT, D>(smem + (ACT ? (i >= NFH ? cb_stride : 0u) : m * cb_stride), idx, shift, lane_off);
                    }
                    // additive codebooks: the weight is the act-dtype-rounded sum of the entries, as the
                    // reference materialises it (embedding_bag output aqlm.py:60-64; __hadd2 in
                    // gemm_kernels.cu:169)
                    a = m == 0 ? am : Act<AT>::add8(a, am);
                }
#pragma unroll
                for (int j = 0; j < TF; ++j) {
                    if constexpr (ABL & 2) {
                        const u32x4 au = __builtin_bit_cast(u32x4, a), xu = __builtin_bit_cast(u32x4, xs.f[s][j]);
                        acc[i][j].x = __uint_as_float(__float_as_uint(acc[i][j].x) ^ au.x ^ au.y ^ xu.x);
                    } else {
                        acc[i][j] = Act<AT>::mfma16(a, xs.f[s][j], acc[i][j]);
                    }
                }
            }
        }
    };

    // ---- everything the first chunk needs is requested before anything is waited for: the codebook
    // entries (needed first), the wave's first codes (HBM, the long latency), its first activations (L2)
    if constexpr (ABL & 64) if (tid == 0) stamps[6] = wall_clock64();   // tile found, addresses set up
    CodebookImage<D, ST_THREADS> img;
    CodebookImage<D, ACT ? ST_THREADS : 1 << 20> img_up;   // ACT: the up partition's codebook, a second full image
    if constexpr (!(ABL & 16))
        img.load(reinterpret_cast<const unsigned char *>(codebooks) + (size_t)loc.part * M * K * EB, M * K, R, tid);
    if constexpr (ACT) img_up.load(reinterpret_cast<const unsigned char *>(codebooks) + (size_t)K * EB, K, R, tid);
    Codes cA, cB;
    XSub X[2];
    if (c_begin + wave < c_end) {
        load_codes(cA, c_begin + wave);
        load_x(X[0], c_begin + wave, 0);
    }
    if constexpr (ABL & 64) if (tid == 0) stamps[7] = wall_clock64();   // image source, codes, activations requested
    // ---- epilogue operands: this thread's output rows in each reduction round.  Fetched now, behind the
    // streaming loads and without touching the loaded registers, so that no global load sits on the tail.
    // round r, thread -> fragment fr = r*RF + tid/64 = (i, j) = (fr / TF, fr % TF); lane l of it holds
    // token 16 j + (l & 15) and rows 16 i + 4 (l >> 4) .. + 3
    uint16_t sc_raw[ROUNDS][4], bi_raw[ROUNDS][4];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int i = (r * RF + (tid >> 6)) / TF;
        const int nb = loc.row0 + 16 * i + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nb + e < row_end ? nb + e : row_end - 1;
            sc_raw[r][e] = scales[n];
            bi_raw[r][e] = bias ? bias[n] : (uint16_t)0;
        }
    }
    uint16_t sc_up[4] = {0, 0, 0, 0}, bi_up[4] = {0, 0, 0, 0};   // ACT: the same for the up rows of the thread's fragment
    if constexpr (ACT) {
        const int nb = loc.row0 + 16 * ((tid >> 6) % NFH) + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = (nb + e < row_end ? nb + e : row_end - 1) + act_rows;
            sc_up[e] = scales[n];
            bi_up[e] = bias ? bias[n] : (uint16_t)0;
        }
    }
    if constexpr (ABL & 64) if (tid == 0) stamps[1] = wall_clock64();   // loads issued
    if constexpr (!(ABL & 16)) img.store(smem, M * K, R, tid);
    if constexpr (ACT) img_up.store(smem + cb_stride, K, R, tid);
    __syncthreads();
    if constexpr (ABL & 64) if (tid == 0) stamps[2] = wall_clock64();   // image built

    // ---- chunks wave, wave + 8, ...: two per iteration so that every register set has a fixed name.
    // The next sub-chunk's activations and the next chunk's codes are in flight while one is computed.
    for (int c = c_begin + wave; c < c_end; c += 2 * ST_WAVES) {
        const int c1 = c + ST_WAVES, c2 = c + 2 * ST_WAVES;
        const bool m1 = c1 < c_end, m2 = c2 < c_end;
        if (m1) load_codes(cB, c1);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if (sub + 1 < NSUB) load_x(X[(sub + 1) & 1], c, sub + 1);
            else if (m1) load_x(X[(sub + 1) & 1], c1, 0);
            place_x(X[sub & 1]);
            compute_sub(cA, X[sub & 1], sub);
        }
        if (m1) {
            if (m2) load_codes(cA, c2);
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                if (sub + 1 < NSUB) load_x(X[(NSUB + sub + 1) & 1], c1, sub + 1);
                else if (m2) load_x(X[(NSUB + sub + 1) & 1], c2, 0);
                place_x(X[(NSUB + sub) & 1]);
                compute_sub(cB, X[(NSUB + sub) & 1], sub);
            }
        }
    }

    if constexpr (ABL & 32) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < TF; ++j) v += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
        if (v == 123.456f) y[tid] = 1;    // keeps the accumulators live, practically never stores
        return;
    }

    if constexpr (ABL & 64) {
        asm volatile("s_nop 0" :: "v"(acc[0][0].x));
        if (tid == 0) stamps[3] = wall_clock64();   // this wave's MFMAs done
    }
    // ---- deterministic cross-wave sum through LDS (the codebook image is dead now), RF fragments a round
    // scale, bias, round to the act dtype, store: thread-private (token t, rows nb .. nb+3) of round r
    auto finish = [&](int r, const f32x4 &sum, int t, int nb) {
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            h[e] = Act<AT>::from_f32(sum[e] * Act<AT>::to_f32(sc_raw[r][e]) + Act<AT>::to_f32(bi_raw[r][e]));
        uint16_t *yp = y + (size_t)t * y_stride + nb;
        if (nb + 4 <= row_end && (reinterpret_cast<uintptr_t>(yp) & 7) == 0) {
            u32x2 o2;
            o2.x = h[0] | ((uint32_t)h[1] << 16);
            o2.y = h[2] | ((uint32_t)h[3] << 16);
            *reinterpret_cast<u32x2 *>(yp) = o2;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nb + e < row_end) yp[e] = h[e];
        }
    };
    const bool split = ksplit > 1;
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);   // [wave][RF][64]
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        __syncthreads();
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            const int fr = r * RF + f;
            red[(wave * RF + f) * 64 + lane] = acc[fr / TF][fr % TF];
        }
        __syncthreads();
        if constexpr (ABL & 64) if (tid == 0 && r == 0) stamps[4] = wall_clock64();   // all waves done
        if constexpr (ACT) {
            if (tid < NFH * 64) {
                const int f = tid >> 6;
                f32x4 g = red[f * 64 + lane], u = red[(f + NFH) * 64 + lane];
#pragma unroll
                for (int w = 1; w < ST_WAVES; ++w) {
                    const f32x4 vg = red[(w * RF + f) * 64 + lane], vu = red[(w * RF + f + NFH) * 64 + lane];
                    g.x += vg.x; g.y += vg.y; g.z += vg.z; g.w += vg.w;
                    u.x += vu.x; u.y += vu.y; u.z += vu.z; u.w += vu.w;
                }
                const int t = t0 + r16;
                const int nb = loc.row0 + 16 * f + q * 4;
                if (t < T && nb < row_end) {
                    uint16_t h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // the two GEMM outputs as the unfused kernel rounds them, then silu_and_mul's two roundings
                        const float yg = Act<AT>::to_f32(Act<AT>::from_f32(g[e] * Act<AT>::to_f32(sc_raw[0][e]) + Act<AT>::to_f32(bi_raw[0][e])));
                        const float yu = Act<AT>::to_f32(Act<AT>::from_f32(u[e] * Act<AT>::to_f32(sc_up[e]) + Act<AT>::to_f32(bi_up[e])));
                        const float sg = Act<AT>::to_f32(Act<AT>::from_f32(yg / (1.0f + expf(-yg))));
                        h[e] = Act<AT>::from_f32(sg * yu);
                    }
                    uint16_t *yp = y + (size_t)t * y_stride + nb;
                    if (nb + 4 <= row_end && (reinterpret_cast<uintptr_t>(yp) & 7) == 0) {
                        u32x2 o2;
                        o2.x = h[0] | ((uint32_t)h[1] << 16);
                        o2.y = h[2] | ((uint32_t)h[3] << 16);
                        *reinterpret_cast<u32x2 *>(yp) = o2;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb + e < row_end) yp[e] = h[e];
                    }
                }
            }
        } else if (tid < RF * 64) {
            const int f = tid >> 6;
            f32x4 sum = red[f * 64 + lane];
#pragma unroll
            for (int w = 1; w < ST_WAVES; ++w) {
                const f32x4 v = red[(w * RF + f) * 64 + lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            // accumulator map: column (token) = lane & 15, row = (lane >> 4) * 4 + reg
            const int fr = r * RF + f;
            const int t = t0 + 16 * (fr % TF) + r16;
            const int nb = loc.row0 + 16 * (fr / TF) + q * 4;
            if (t < T && nb < row_end) {
                if (!split) finish(r, sum, t, nb);
                else coherent_store16(ws_part + ((size_t)zsplit * T + t) * (size_t)tmap.row_end[PQ_MAX_PARTS - 1] + nb, sum);
            }
        }
    }
    // ---- K split over blocks: partial sums [ksplit][T][N] fp32 in the workspace; every block counts itself
    // in on its tile's counter and the last one to arrive adds the partials up in split order (so the result
    // does not depend on which block that is), finishes the tile and leaves the counter at zero again.
    // Partials and counter are written and read with device-scope relaxed atomics (coherent accesses that
    // go past the per-XCD L2), ordered by waiting for the stores' completion before the count: a device-
    // scope fence would write back and invalidate the whole L2 instead, tens of microseconds per launch.
    if (split) {
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partials have been written ...
        __syncthreads();                                    // ... and so have the block's
        if (tid == 0) {
            unsigned *cnt = ws_cnt + (size_t)blockIdx.y * n_tiles_x + tile_x;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)ksplit - 1u;
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last) {
            const size_t n_total = (size_t)tmap.row_end[PQ_MAX_PARTS - 1];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                if (tid < RF * 64) {
                    const int fr = r * RF + (tid >> 6);
                    const int t = t0 + 16 * (fr % TF) + r16;
                    const int nb = loc.row0 + 16 * (fr / TF) + q * 4;
                    if (t < T && nb < row_end) {
                        f32x4 part[8];   // all partials requested before any is waited for (ksplit <= 8)
#pragma unroll
                        for (int z = 0; z < 8; ++z) {
                            // no branch between issue and wait (a join could make the compiler copy a register whose load is
                            // still in flight): splits past the last re-read the last one and are left out of the sum
                            const int zc = z < ksplit ? z : ksplit - 1;
                            part[z] = coherent_load16_issue(ws_part + ((size_t)zc * T + t) * n_total + nb);
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        f32x4 sum{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int z = 0; z < 8; ++z) coherent_landed(part[z]);
#pragma unroll
                        for (int z = 0; z < 8; ++z) {
                            if (z < ksplit) { sum.x += part[z].x; sum.y += part[z].y; sum.z += part[z].z; sum.w += part[z].w; }
                        }
                        finish(r, sum, t, nb);
                    }
                }
            }
        }
    }
    if constexpr (ABL & 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) stamps[5] = wall_clock64();   // stores retired
    }
}

// ---- per-dtype launcher (instantiated once per act dtype, in its own translation unit) --------
template <typename AT, typename CodeT, int D, int M, int NF, int TF, bool PACKED>
static hipError_t launch_nf(const GemmArgs &a, int ksplit, hipStream_t stream) {
    const int R = stream_replication<D>(a.K, a.M);
    const int nbits = __builtin_ctz((unsigned)a.K);
    constexpr int RF = stream_reduce_frags(NF * TF);
    size_t lds = (size_t)a.M * a.K * 2 * D * R;
    const size_t red = (size_t)ST_WAVES * RF * 64 * 16;
    if (red > lds) lds = red;
    typedef void (*kern_t)(const uint16_t *, int, int64_t, const void *, const uint16_t *, const uint16_t *, const uint16_t *,
                           TileMap, int, int, int, int, uint16_t *, int64_t, int, float *, unsigned *, int);
    kern_t kern = gemm_stream_kernel<AT, CodeT, D, M, NF, TF, PACKED>;
    if constexpr (PACKED) {
        if (a.T >= 16) {
            // lanes along the rows + placement through the waves' activation slots (a KiB per operand register in
            // flight: 8 KiB per wave) behind image and reduction area
            kern = gemm_stream_kernel<AT, CodeT, D, M, NF, TF, PACKED, 0, false, true>;
            constexpr int KS_ = StreamFmt<sizeof(CodeT), D, M>::KS, SUBK_ = (8 / TF) < KS_ ? (8 / TF) : KS_;
            lds = stream_x_lds_offset((size_t)a.M * a.K * 2 * D * R, red) + (size_t)ST_WAVES * SUBK_ * TF * 1024;
            static std::atomic<bool> attr_set[64];   // per device: more than the default 64 KiB of dynamic LDS
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
            if (!attr_set[dev].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(2 * ST_MAX_LDS_CB));
                if (e != hipSuccess) return e;
                attr_set[dev].store(true, std::memory_order_release);
            }
        }
    }
    const int n_tiles = tiles_in(a.parts, NF * 16);
    const dim3 grid(n_tiles * ksplit, (a.T + 16 * TF - 1) / (16 * TF));
    unsigned *ws_cnt = static_cast<unsigned *>(a.workspace);
    float *ws_part = reinterpret_cast<float *>(static_cast<unsigned char *>(a.workspace) + WS_COUNTER_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(ST_THREADS), lds, stream,
                       static_cast<const uint16_t *>(a.x), a.T, a.x_stride, a.codes,
                       static_cast<const uint16_t *>(a.codebooks), static_cast<const uint16_t *>(a.scales),
                       static_cast<const uint16_t *>(a.bias), make_tile_map(a.parts, NF * 16), a.G, a.K, nbits, R,
                       static_cast<uint16_t *>(a.y), a.y_stride, ksplit, ws_part, ws_cnt, 0);
    return hipGetLastError();
}

// Fused gate/up GEMM + SiLU-and-mul (d = 8, one codebook, one-byte codes, packed): a.parts holds the two equal
// partitions, a.y is [T][N / 2].
template <typename AT, int NF>
static hipError_t launch_act_nf(const GemmArgs &a, hipStream_t stream) {
    const int rows = a.parts.size[0];
    const int R = stream_replication<8>(a.K, 1);            // a full image per codebook: 2 x 64 KiB at K = 256
    const int nbits = __builtin_ctz((unsigned)a.K);
    size_t lds = (size_t)2 * a.K * 16 * R;
    const size_t red = (size_t)ST_WAVES * NF * 64 * 16;
    if (red > lds) lds = red;
    Parts gate;
    gate.num = 1;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) gate.size[p] = p == 0 ? rows : 0;
    const dim3 grid(tiles_in(gate, NF * 8), (a.T + 15) / 16);
    auto kern = gemm_stream_kernel<AT, int8_t, 8, 1, NF, 1, true, 0, true>;
    static std::atomic<bool> attr_set[64];   // per device: the kernel may use more than the default 64 KiB of LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * ST_MAX_LDS_CB);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, dim3(ST_THREADS), lds, stream,
                       static_cast<const uint16_t *>(a.x), a.T, a.x_stride, a.codes,
                       static_cast<const uint16_t *>(a.codebooks), static_cast<const uint16_t *>(a.scales),
                       static_cast<const uint16_t *>(a.bias), make_tile_map(gate, NF * 8), a.G, a.K, nbits, R,
                       static_cast<uint16_t *>(a.y), a.y_stride, 1, nullptr, nullptr, rows);
    return hipGetLastError();
}

template <typename AT>
static hipError_t launch_act_at(const GemmArgs &a, hipStream_t stream) {
    switch (choose_stream_act_nf(a)) {
    case 8: return launch_act_nf<AT, 8>(a, stream);
    case 4: return launch_act_nf<AT, 4>(a, stream);
    default: return launch_act_nf<AT, 2>(a, stream);
    }
}

template <typename AT, typename CodeT, int D, int M>
static hipError_t launch_fmt(const GemmArgs &a, hipStream_t stream) {
    constexpr bool LIGHT = D == 8 && M == 1 && sizeof(CodeT) == 1;   // == stream_light_format(a)
    const StreamCfg c = choose_stream_cfg(a, nullptr);
    if (a.packed) {
        if (c.tf == 4) {
            if constexpr (LIGHT)
                if (c.nf == 4) return launch_nf<AT, CodeT, D, M, 4, 4, true>(a, c.ksplit, stream);
            return launch_nf<AT, CodeT, D, M, 2, 4, true>(a, c.ksplit, stream);
        }
        if (c.tf == 2) {
            if constexpr (LIGHT)
                if (c.nf == 7) return launch_nf<AT, CodeT, D, M, 7, 2, true>(a, c.ksplit, stream);
            return launch_nf<AT, CodeT, D, M, 4, 2, true>(a, c.ksplit, stream);
        }
        if constexpr (LIGHT) {
            switch (c.nf) {
            case 8: return launch_nf<AT, CodeT, D, M, 8, 1, true>(a, c.ksplit, stream);
            case 7: return launch_nf<AT, CodeT, D, M, 7, 1, true>(a, c.ksplit, stream);
            case 4: return launch_nf<AT, CodeT, D, M, 4, 1, true>(a, c.ksplit, stream);
            default: break;
            }
        }
        return c.nf == 2 ? launch_nf<AT, CodeT, D, M, 2, 1, true>(a, c.ksplit, stream)
                         : launch_nf<AT, CodeT, D, M, 1, 1, true>(a, c.ksplit, stream);
    }
    if constexpr (LIGHT)
        if (c.nf == 4) return launch_nf<AT, CodeT, D, M, 4, 1, false>(a, c.ksplit, stream);
    return c.nf == 2 ? launch_nf<AT, CodeT, D, M, 2, 1, false>(a, c.ksplit, stream)
                     : launch_nf<AT, CodeT, D, M, 1, 1, false>(a, c.ksplit, stream);
}

template <typename AT>
static hipError_t launch_at(const GemmArgs &a, hipStream_t stream) {
    if (a.M == 2) return launch_fmt<AT, int8_t, 8, 2>(a, stream);
    if (a.code_bytes == 1) {
        switch (a.d) {
        case 8: return launch_fmt<AT, int8_t, 8, 1>(a, stream);
        case 4: return launch_fmt<AT, int8_t, 4, 1>(a, stream);
        default: return launch_fmt<AT, int8_t, 2, 1>(a, stream);
        }
    }
    switch (a.d) {
    case 8: return launch_fmt<AT, int16_t, 8, 1>(a, stream);
    case 4: return launch_fmt<AT, int16_t, 4, 1>(a, stream);
    default: return launch_fmt<AT, int16_t, 2, 1>(a, stream);
    }
}

}  // namespace pq
