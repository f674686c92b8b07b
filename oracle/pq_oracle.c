/*
 * pq_oracle.c -- CPU restatement of the reference's PQ / AQLM linear path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker that the HIP kernels are compared
 * against (tests/, __graft_entry__.smoke(), and the cpu_baseline leg of bench.py).  It is
 * never linked into, imported by, or called from the product library (libpq_amd.so) or
 * the vllm_pq_amd Python package.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors that were
 * produced by running the reference's own Python functions (tests/golden/make_golden.py,
 * tests/test_oracle_golden.py): bit-exact for index decoding, the gather and the scaled
 * dequantised weight; within a stated tolerance for the GEMM (the reference's CPU GEMM
 * uses a different summation order).
 *
 * Reference functions restated (file = vllm/model_executor/layers/quantization/aqlm.py):
 *   unpack_int_data            aqlm.py:32-34    -> pq_oracle_unpack
 *   dequantize_weight          aqlm.py:37-74    -> pq_oracle_dequant (one partition at a time)
 *   dequantize_gemm            aqlm.py:77-90    -> pq_oracle_gemm, order 0
 *   generic_dequantize_gemm    aqlm.py:94-127   -> partition loop in both functions
 *   optimized_dequantize_gemm  aqlm.py:132-156  -> pq_oracle_gemm, order 1 (GEMM, then scale the
 *                                                  rounded output; bias path scales W first)
 *
 * Tensors are plain row-major arrays.  fp16 / bf16 values travel as uint16 bit patterns.
 *   codes      [N][G][M]   int8 / int16 / int32 (code_bytes = 1, 2, 4), signed storage
 *   codebooks  [P*M][K][d] act dtype   (out_group_size == 1, aqlm.py:179)
 *   scales     [N]         act dtype   (reference shape [N,1,1,1])
 *   x          [T][Kin]    act dtype,  Kin = G*d
 *   y          [T][N]      act dtype
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define PQ_FP16 0
#define PQ_BF16 1

/* ---- scalar fp16 / bf16 <-> fp32, round-to-nearest-even ---------------------------- */
static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float fp16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f32_to_fp16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);      /* NaN */
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);     /* >= 65536 (or inf) -> inf */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);     /* rounds up to inf */
    if (a < 0x33000001u) return (uint16_t)sign;                  /* < 2^-25 (incl. tie) -> 0 */
    int e = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out;
    if (hexp == 0) out = q;                     /* may carry into exponent 1: correct */
    else out = ((hexp - 1u) << 10) + q;         /* q has the implicit bit at 0x400 */
    return (uint16_t)(sign | out);
}
static inline float act_to_f32(uint16_t h, int dtype) { return dtype == PQ_BF16 ? bf16_to_f32(h) : fp16_to_f32(h); }
static inline uint16_t f32_to_act(float f, int dtype) { return dtype == PQ_BF16 ? f32_to_bf16(f) : f32_to_fp16(f); }

/*
 * How the fp32 sum over M >= 2 codebooks is rounded to bf16.
 *   0: round-to-nearest-even (what a GPU __hadd2-style sum and this build's HIP kernels do).
 *   1: add 0x8000 and truncate (ties away from zero).  This is what torch's CPU embedding_bag
 *      (the call at aqlm.py:60-64) does for bf16 tables in torch 2.10, found by comparing with
 *      the golden vectors; it only matters on exact ties.  fp16 tables and M == 1 are unaffected.
 */
static int g_bf16_sum_mode = 0;
void pq_oracle_set_bf16_sum_rounding(int mode) { g_bf16_sum_mode = mode; }
static uint16_t sum_to_act(float f, int dtype) {
    if (dtype == PQ_BF16 && g_bf16_sum_mode == 1) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
        return (uint16_t)((u + 0x8000u) >> 16);
    }
    return f32_to_act(f, dtype);
}

/* exported for the conversion self-test */
float pq_oracle_act_to_f32(uint16_t h, int dtype) { return act_to_f32(h, dtype); }
uint16_t pq_oracle_f32_to_act(float f, int dtype) { return f32_to_act(f, dtype); }

/* ---- aqlm.py:32-34  unpack_int_data: data.to(int64) % 2**nbits -------------------- */
static inline int64_t load_code(const void *codes, int code_bytes, int64_t i) {
    switch (code_bytes) {
    case 1: return ((const int8_t *)codes)[i];
    case 2: return ((const int16_t *)codes)[i];
    default: return ((const int32_t *)codes)[i];
    }
}
static inline int64_t pymod(int64_t a, int64_t m) { /* Python / torch remainder: sign of divisor */
    int64_t r = a % m;
    return r < 0 ? r + m : r;
}
int pq_oracle_unpack(const void *codes, int code_bytes, int64_t n, int nbits, int64_t *out) {
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4) return -1;
    if (nbits < 1 || nbits > 32) return -1;
    const int64_t K = (int64_t)1 << nbits;
    for (int64_t i = 0; i < n; i++) out[i] = pymod(load_code(codes, code_bytes, i), K);
    return 0;
}

/* nbits as the reference derives it: codebooks.shape[1].bit_length() - 1  (aqlm.py:86) */
static int nbits_of(int K) {
    int b = 0;
    while ((K >> b) > 1) b++;
    return b;
}

/*
 * aqlm.py:37-74 dequantize_weight + the partition loop of aqlm.py:105-126.
 *   W[n, g*d + j] = act( act( sum_m cb[p(n)*M + m][idx(n,g,m)][j] ) * scale[n] )
 * The sum over m is formed in fp32 and rounded once (embedding_bag mode="sum"); the scale multiply
 * is one more rounding (Tensor.mul in the act dtype).  scales == NULL -> unscaled (pure gather when M=1).
 * Wf (optional) receives the same values widened to fp32, for the GEMM below.
 */
int pq_oracle_dequant(const void *codes, int code_bytes, const uint16_t *codebooks, const uint16_t *scales,
                      const int *parts, int P, int N, int G, int M, int K, int d, int dtype,
                      uint16_t *W, float *Wf) {
    if (code_bytes != 1 && code_bytes != 2 && code_bytes != 4) return -1;
    if (K < 2 || (K & (K - 1))) return -2; /* reference needs a power of two (aqlm.py:86) */
    int64_t tot = 0;
    for (int p = 0; p < P; p++) tot += parts[p];
    if (tot != N) return -3; /* aqlm.py:112 */
    const int nbits = nbits_of(K);
    const int64_t Kmod = (int64_t)1 << nbits;
    const int Kin = G * d;
    int *row_part = (int *)malloc(sizeof(int) * (size_t)N);
    if (!row_part) return -4;
    {
        int n = 0;
        for (int p = 0; p < P; p++)
            for (int i = 0; i < parts[p]; i++) row_part[n++] = p;
    }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        const int p = row_part[n];
        const float s = scales ? act_to_f32(scales[n], dtype) : 1.0f;
        for (int g = 0; g < G; g++) {
            for (int j = 0; j < d; j++) {
                float acc = 0.f;
                for (int m = 0; m < M; m++) {
                    int64_t idx = pymod(load_code(codes, code_bytes, ((int64_t)n * G + g) * M + m), Kmod);
                    acc += act_to_f32(codebooks[(((int64_t)(p * M + m)) * K + idx) * d + j], dtype);
                }
                uint16_t h;
                if (M == 1) {
                    /* a single gathered entry passes through unchanged (bit pattern preserved) */
                    int64_t idx = pymod(load_code(codes, code_bytes, ((int64_t)n * G + g) * M), Kmod);
                    h = codebooks[(((int64_t)(p * M)) * K + idx) * d + j];
                } else {
                    h = sum_to_act(acc, dtype);
                }
                if (scales) h = f32_to_act(act_to_f32(h, dtype) * s, dtype);
                if (W) W[(int64_t)n * Kin + g * d + j] = h;
                if (Wf) Wf[(int64_t)n * Kin + g * d + j] = act_to_f32(h, dtype);
            }
        }
    }
    free(row_part);
    return 0;
}

/*
 * y = x @ W^T (+ bias), fp32 (or fp64) accumulation, one rounding of the result.
 *   order 0: aqlm.py:77-90  -- W scaled and rounded to the act dtype first ("scale the weight").
 *   order 1: aqlm.py:143-151 -- unscaled W, y = act( act(x @ W^T) * scale ) when bias == NULL;
 *            with a bias the reference scales W first (aqlm.py:153-156), i.e. order 0.
 *   order 2: not a reference order: y = act( scale * (x @ W_unscaled^T) + bias ) evaluated in the
 *            accumulator precision -- the exact-arithmetic value both reference orders approximate.
 * y32 (optional) receives the un-rounded result.
 */
int pq_oracle_gemm(const uint16_t *x, int T, const void *codes, int code_bytes, const uint16_t *codebooks,
                   const uint16_t *scales, const uint16_t *bias, const int *parts, int P, int N, int G, int M,
                   int K, int d, int dtype, int order, int acc_double, uint16_t *y, float *y32) {
    const int Kin = G * d;
    if (order == 1 && bias) order = 0;
    float *Wf = (float *)malloc(sizeof(float) * (size_t)N * Kin);
    float *xf = (float *)malloc(sizeof(float) * (size_t)T * Kin);
    if (!Wf || !xf) { free(Wf); free(xf); return -4; }
    int rc = pq_oracle_dequant(codes, code_bytes, codebooks, order == 0 ? scales : NULL, parts, P, N, G, M, K, d,
                               dtype, NULL, Wf);
    if (rc) { free(Wf); free(xf); return rc; }
    for (int64_t i = 0; i < (int64_t)T * Kin; i++) xf[i] = act_to_f32(x[i], dtype);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        const float *w = Wf + (int64_t)n * Kin;
        const float s = act_to_f32(scales[n], dtype);
        const float b = bias ? act_to_f32(bias[n], dtype) : 0.f;
        for (int t = 0; t < T; t++) {
            const float *xr = xf + (int64_t)t * Kin;
            double r;
            if (acc_double) {
                double a = 0.0;
                for (int k = 0; k < Kin; k++) a += (double)xr[k] * (double)w[k];
                r = a;
            } else {
                float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int k = 0;
                for (; k + 8 <= Kin; k += 8)
                    for (int j = 0; j < 8; j++) a[j] += xr[k + j] * w[k + j];
                float tail = 0.f;
                for (; k < Kin; k++) tail += xr[k] * w[k];
                r = (double)(((a[0] + a[4]) + (a[1] + a[5])) + ((a[2] + a[6]) + (a[3] + a[7])) + tail);
            }
            float out;
            if (order == 0) {
                out = (float)(r + (double)b);
            } else if (order == 1) {
                out = act_to_f32(f32_to_act((float)r, dtype), dtype) * s; /* bias == NULL here */
            } else {
                out = (float)(r * (double)s + (double)b);
            }
            if (y) y[(int64_t)t * N + n] = f32_to_act(out, dtype);
            if (y32) y32[(int64_t)t * N + n] = out;
        }
    }
    free(Wf);
    free(xf);
    return 0;
}

int pq_oracle_version(void) { return 1; }
