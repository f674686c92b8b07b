// Shared device helpers for the gfx950 PQ kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pq_amd.h"

namespace pq {

typedef __bf16 bf16_t;
typedef _Float16 fp16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fp16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Output-partition table, passed by value.  Rows [start[p], start[p] + size[p]) use codebooks
// p*M .. p*M+M-1  (generic_dequantize_gemm's narrow() calls, aqlm.py:115-126).
struct Parts {
    int num;
    int size[PQ_MAX_PARTS];
};

// Maps a tile index to (partition, first row, row count) when every partition is cut into tiles of
// `rows_per_tile` rows separately, so that no tile straddles two codebooks.
struct TileLoc {
    int part;
    int row0;   // absolute first row
    int nrows;  // rows in this tile (<= rows_per_tile)
};

__host__ __device__ inline int tiles_in(const Parts &P, int rows_per_tile) {
    int t = 0;
    for (int p = 0; p < P.num; ++p) t += (P.size[p] + rows_per_tile - 1) / rows_per_tile;
    return t;
}

__device__ inline TileLoc locate_tile(const Parts &P, int tile, int rows_per_tile) {
    TileLoc L;
    int row = 0;
    L.part = 0;
    L.row0 = 0;
    L.nrows = 0;
#pragma unroll
    for (int p = 0; p < PQ_MAX_PARTS; ++p) {
        if (p < P.num) {
            const int nt = (P.size[p] + rows_per_tile - 1) / rows_per_tile;
            if (tile >= 0 && tile < nt) {
                L.part = p;
                L.row0 = row + tile * rows_per_tile;
                const int left = P.size[p] - tile * rows_per_tile;
                L.nrows = left < rows_per_tile ? left : rows_per_tile;
                tile = -1;
            } else if (tile >= nt) {
                tile -= nt;
            }
            row += P.size[p];
        }
    }
    return L;
}

// The same map with the per-partition prefix sums worked out on the host, so that a block finds its
// tile with a handful of scalar compare/selects instead of a division loop over the partitions (that
// loop is ~300 dependent scalar instructions, over a microsecond at the head of a short kernel).
struct TileMap {
    int tile0[PQ_MAX_PARTS];    // first tile of partition p (INT_MAX for p >= num)
    int row0[PQ_MAX_PARTS];     // first row of partition p
    int row_end[PQ_MAX_PARTS];  // one past its last row
};

// Rows [lo, hi) of the layer only (a launch over part of the rows: every partition clipped to the range, its tiles
// counted from the clipped start; a partition outside the range has no tiles and is never selected).
inline TileMap make_tile_map_range(const Parts &P, int rows_per_tile, int lo, int hi) {
    TileMap m;
    int tile = 0, row = 0;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) {
        if (p < P.num) {
            const int a = row > lo ? row : lo, b = row + P.size[p] < hi ? row + P.size[p] : hi;
            m.tile0[p] = tile;
            m.row0[p] = a < b ? a : row;
            m.row_end[p] = a < b ? b : row;
            if (a < b) tile += (b - a + rows_per_tile - 1) / rows_per_tile;
            row += P.size[p];
        } else {
            m.tile0[p] = 0x7fffffff;
            m.row0[p] = row;
            m.row_end[p] = row;
        }
    }
    return m;
}

inline int tiles_in_range(const Parts &P, int rows_per_tile, int lo, int hi) {
    int t = 0, row = 0;
    for (int p = 0; p < P.num; ++p) {
        const int a = row > lo ? row : lo, b = row + P.size[p] < hi ? row + P.size[p] : hi;
        if (a < b) t += (b - a + rows_per_tile - 1) / rows_per_tile;
        row += P.size[p];
    }
    return t;
}

inline TileMap make_tile_map(const Parts &P, int rows_per_tile) {
    return make_tile_map_range(P, rows_per_tile, 0, 0x7fffffff);
}

__device__ inline TileLoc locate_tile(const TileMap &m, int tile, int rows_per_tile) {
    int part = 0, t0 = m.tile0[0], r0 = m.row0[0], re = m.row_end[0];
#pragma unroll
    for (int p = 1; p < PQ_MAX_PARTS; ++p) {
        const bool ge = tile >= m.tile0[p];
        part = ge ? p : part;
        t0 = ge ? m.tile0[p] : t0;
        r0 = ge ? m.row0[p] : r0;
        re = ge ? m.row_end[p] : re;
    }
    TileLoc L;
    L.part = part;
    L.row0 = r0 + (tile - t0) * rows_per_tile;
    const int left = re - L.row0;
    L.nrows = left < rows_per_tile ? left : rows_per_tile;
    return L;
}


// ---- act dtype traits ---------------------------------------------------------------------
template <typename AT>
struct Act;
template <>
struct Act<bf16_t> {
    typedef bf16x8 vec8;
    static __device__ inline float to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
    static __device__ inline uint16_t from_f32(float f) {
        bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: nearest even, NaN stays NaN
        return __builtin_bit_cast(uint16_t, b);
    }
    // two values rounded and packed, a in the low half: one v_cvt_pk_bf16_f32
    static __device__ inline uint32_t pack2(float a, float b) {
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 h = {(bf16_t)a, (bf16_t)b};
        return __builtin_bit_cast(uint32_t, h);
    }
    // element-wise a + b, each sum rounded once to bf16 (nearest even).  gfx950 has no packed bf16 add: per pair of
    // values two unpacks each, ONE packed fp32 add (written on float pairs so that the compiler emits v_pk_add_f32)
    // and one packed conversion -- 6 instructions instead of 7
    static __device__ inline vec8 add8(vec8 a, vec8 b) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
        u32x4 r;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2 fa = {__uint_as_float(ua[p] << 16), __uint_as_float(ua[p] & 0xffff0000u)};
            const f32x2 fb = {__uint_as_float(ub[p] << 16), __uint_as_float(ub[p] & 0xffff0000u)};
            const f32x2 sum = fa + fb;
            const bf16x2 h = {(bf16_t)sum.x, (bf16_t)sum.y};
            r[p] = __builtin_bit_cast(uint32_t, h);
        }
        return __builtin_bit_cast(vec8, r);
    }
    static __device__ inline f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ inline f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Act<fp16_t> {
    typedef fp16x8 vec8;
    static __device__ inline float to_f32(uint16_t h) { return (float)__builtin_bit_cast(fp16_t, h); }
    static __device__ inline uint16_t from_f32(float f) {
        fp16_t b = (fp16_t)f;
        return __builtin_bit_cast(uint16_t, b);
    }
    static __device__ inline uint32_t pack2(float a, float b) {
        typedef fp16_t fp16x2 __attribute__((ext_vector_type(2)));
        const fp16x2 h = {(fp16_t)a, (fp16_t)b};
        return __builtin_bit_cast(uint32_t, h);
    }
    static __device__ inline vec8 add8(vec8 a, vec8 b) { return a + b; }   // v_pk_add_f16: correctly rounded
    static __device__ inline f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ inline f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// An fp32 value as it stands in a register.  The compiler may otherwise merge the arithmetic that produced it with a
// following conversion to fp16 into one v_fma_mix instruction, which rounds the exact result ONCE; where the reference
// rounds to fp32 first and to the act dtype second (RMSNorm.forward_native, layernorm.py:66-70), the two can differ
// in the last bit.
__device__ __forceinline__ float rounded_f32(float f) {
    asm("" : "+v"(f));
    return f;
}

// Unsigned index of a signed stored code: value mod K with K a power of two (aqlm.py:32-34).
template <typename CodeT>
__device__ inline uint32_t code_index(CodeT c, uint32_t kmask) {
    return (uint32_t)(int32_t)c & kmask;
}


// Device-coherent 16-byte accesses (sc1: past the per-XCD L2, what a device-scope relaxed atomic access
// compiles to) for data that blocks on different XCDs hand to each other inside one launch.  The load is only
// issued: several can be in flight; `s_waitcnt vmcnt(0)` and coherent_landed() on each value come before use.
__device__ __forceinline__ void coherent_store16(float *p, const f32x4 &v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 coherent_load16_issue(const float *p) {
    f32x4 v;
    asm volatile("; PQI %0\n\tglobal_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void coherent_landed(f32x4 &v) { asm volatile("; PQL %0" : "+v"(v)); }

}  // namespace pq
