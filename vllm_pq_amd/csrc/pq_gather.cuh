// Device helpers shared by the gather+MFMA kernels: decoding a lane's codes, reading an MFMA operand
// fragment out of the replicated LDS codebook image, and building that image.
#pragma once
#include "pq_common.cuh"
#include "pq_stream_geom.cuh"

namespace pq {

template <typename AT, int D>
__device__ inline typename Act<AT>::vec8 gather_frag(const unsigned char *cb_lds, const uint32_t (&idx)[8 / D],
                                                      uint32_t shift, uint32_t lane_off) {
    typedef typename Act<AT>::vec8 vec8;
    if constexpr (D == 8) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(cb_lds + ((idx[0] << shift) + lane_off));
        return __builtin_bit_cast(vec8, v);
    } else if constexpr (D == 4) {
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(cb_lds + ((idx[0] << shift) + lane_off));
        const u32x2 hi = *reinterpret_cast<const u32x2 *>(cb_lds + ((idx[1] << shift) + lane_off));
        u32x4 v;
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        return __builtin_bit_cast(vec8, v);
    } else {
        u32x4 v;
        v.x = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[0] << shift) + lane_off));
        v.y = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[1] << shift) + lane_off));
        v.z = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[2] << shift) + lane_off));
        v.w = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[3] << shift) + lane_off));
        return __builtin_bit_cast(vec8, v);
    }
}

// Code `i` of the 16 bytes a lane loaded (CB = bytes per code), reduced mod K = low nbits.
template <int CB, typename VecT>
__device__ inline uint32_t code_at(const VecT &w, int i, uint32_t nbits) {
    if constexpr (CB == 1) {
        return __builtin_amdgcn_ubfe(w[i >> 2], (uint32_t)((i & 3) * 8), nbits);
    } else {
        return __builtin_amdgcn_ubfe(w[i >> 1], (uint32_t)((i & 1) * 16), nbits);
    }
}

// The replicated codebook image of one partition in LDS: 16-byte slot j holds entry j / (EB*R/16).
// Split in two so that the source loads (L2-resident, issued first) are in flight together with the
// kernel's other global loads and the LDS stores wait for them alone.
// When an entry fills at least MAX_IT slots (every format with a replicated image of full size), a thread reads ONE
// entry and writes MAX_IT of its slots: one global load per thread instead of MAX_IT loads of the same bytes --
// the texture path of a CU takes 16 cycles per 16-byte wave-load whatever the addresses, and a short kernel's
// first microseconds are spent pushing its first loads through it.  The slots go out in a per-lane rotated order
// so that the lanes of a store group land on different banks.
template <int D, int THREADS>
struct CodebookImage {
    static constexpr int EB = 2 * D;
    static constexpr int MAX_IT = (ST_MAX_LDS_CB / 16 + THREADS - 1) / THREADS;
    u32x4 v[MAX_IT];

    static __device__ inline bool one_load(int R) { return ((EB * R) / 16) % MAX_IT == 0; }

    __device__ inline u32x4 fetch(const unsigned char *cb_g, int e) const {
        if constexpr (D == 8) {
            return *reinterpret_cast<const u32x4 *>(cb_g + (size_t)e * 16);
        } else if constexpr (D == 4) {
            const u32x2 s2 = *reinterpret_cast<const u32x2 *>(cb_g + (size_t)e * 8);
            return u32x4{s2.x, s2.y, s2.x, s2.y};
        } else {
            const uint32_t s1 = *reinterpret_cast<const uint32_t *>(cb_g + (size_t)e * 4);
            return u32x4{s1, s1, s1, s1};
        }
    }

    __device__ inline void load(const unsigned char *cb_g, int entries, int R, int tid) {
        const int slots = (entries * EB * R) / 16;
        const int spe = (EB * R) / 16;                 // slots per entry
        if (one_load(R)) {
            const int e = tid / (spe / MAX_IT);
            v[0] = fetch(cb_g, e < entries ? e : entries - 1);
            return;
        }
        const int per_entry_shift = __builtin_ctz((unsigned)spe);
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            int j = tid + it * THREADS;
            j = j < slots ? j : slots - 1;     // clamp instead of branching; extra slots are not stored
            v[it] = fetch(cb_g, j >> per_entry_shift);
        }
    }
    __device__ inline void store(unsigned char *smem, int entries, int R, int tid) const {
        const int slots = (entries * EB * R) / 16;
        if (one_load(R)) {
            const int first = tid * MAX_IT;            // this thread's MAX_IT consecutive slots, all of one entry
            if (first < slots) {
#pragma unroll
                for (int c = 0; c < MAX_IT; ++c)
                    *reinterpret_cast<u32x4 *>(smem + (size_t)(first + ((c + tid) & (MAX_IT - 1))) * 16) = v[0];
            }
            return;
        }
        if (slots == MAX_IT * THREADS) {       // a full-size image: no predicates
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it)
                *reinterpret_cast<u32x4 *>(smem + (size_t)(tid + it * THREADS) * 16) = v[it];
        } else {
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it) {
                const int j = tid + it * THREADS;
                if (j < slots) *reinterpret_cast<u32x4 *>(smem + (size_t)j * 16) = v[it];
            }
        }
    }
};

}  // namespace pq
