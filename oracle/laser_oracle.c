/* oracle/laser_oracle.c
 *
 * ===========================================================================================
 *  TEST INFRASTRUCTURE ONLY.  This is the CPU oracle for the laser_amd hot path: a plain-C
 *  restatement of mratsim/laser's packed-panel GEMM (laser/primitives/matrix_multiplication/),
 *  its physical transposes (laser/primitives/swapaxes.nim) and the im2col->GEMM convolution
 *  (benchmarks/convolution/conv2d_im2col.nim).  Only tests/, __graft_entry__.smoke() and
 *  bench.py's `cpu_baseline` leg may load it -- and only as the checker / the timed CPU
 *  baseline, never as the product.  The product path (laser_amd/, liblaser_hip.so) never links,
 *  imports or calls anything in this directory and fails loudly without its HIP library.
 *
 *  PARITY PIN: the reference itself cannot be built here (Nim toolchain absent; its mandatory
 *  pytorch/cpuinfo C sources are an empty un-vendored submodule, .gitmodules:1-3), so there is
 *  no oracle/_ref.  This restatement is pinned instead against EVERY known-answer test the
 *  reference holds for this path: gemm.nim:257-507 (8 KATs), gemm_prepacked.nim:354-523
 *  (9 KATs), benchmarks/convolution/conv2d_common.nim:147-283 (2 conv KATs) -- transcribed in
 *  tests/golden/laser_kats.json and checked by tests/test_oracle_golden.py -- plus float64
 *  naive products and numpy/OpenBLAS (the reference's own "vendor BLAS" comparator).
 * ===========================================================================================
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off; SIMD kernels are selected at
 * RUN time with __builtin_cpu_supports, mirroring the reference's cpuinfo dispatch,
 * gemm.nim:228-247, so one binary is safe on any x86-64 host).
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* CPUFeatureX86 -- gemm_tiling.nim:72-84 (same order) */
enum {
  ISA_GENERIC = 0,
  ISA_SSE = 1,
  ISA_SSE2 = 2,
  ISA_SSE4_1 = 3,
  ISA_AVX = 4,
  ISA_AVX_FMA = 5,
  ISA_AVX2 = 6,
  ISA_AVX512 = 7
};

static int has_avx512(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }
static int has_fma(void) { return __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2"); }

/* Runtime ISA pick per dtype -- gemm.nim:228-247.  dtype: 0 f32, 1 f64, 2 i32, 3 i64. */
int oracle_detect_isa(int dtype) {
  int avx512 = has_avx512(), fma = __builtin_cpu_supports("fma"), avx = __builtin_cpu_supports("avx"),
      avx2 = __builtin_cpu_supports("avx2"), sse41 = __builtin_cpu_supports("sse4.1");
  switch (dtype) {
    case 0: return avx512 ? ISA_AVX512 : fma ? ISA_AVX_FMA : avx ? ISA_AVX : ISA_SSE;
    case 1: return avx512 ? ISA_AVX512 : fma ? ISA_AVX_FMA : avx ? ISA_AVX : ISA_SSE2;
    case 2: return avx512 ? ISA_AVX512 : avx2 ? ISA_AVX2 : sse41 ? ISA_SSE4_1 : ISA_SSE2;
    default: return avx512 ? ISA_AVX512 : ISA_SSE2;
  }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * SIMD micro-kernels for the TIMED baseline (the "Laser OpenMP CPU path"): the same register
 * blocking the reference generates -- gemm_ukernel_generator.nim:140-250: per k, load NbVecs=2
 * aligned vectors of B~, broadcast A~[k*MR+i], AB[i][jj] = fma(a, b[jj], AB[i][jj]).
 * Bitwise equal to the scalar restatement (tests/test_oracle_golden.py checks this).
 * ---------------------------------------------------------------------------------------- */

__attribute__((target("avx512f"))) static void uk_f32_avx512_14x32(int64_t kc, const float *A,
                                                                   const float *B, float *AB) {
  __m512 c[14][2];
  for (int i = 0; i < 14; i++) c[i][0] = c[i][1] = _mm512_setzero_ps();
  for (int64_t k = 0; k < kc; k++) {
    __m512 b0 = _mm512_load_ps(B + k * 32), b1 = _mm512_load_ps(B + k * 32 + 16);
#pragma GCC unroll 14
    for (int i = 0; i < 14; i++) {
      __m512 a = _mm512_set1_ps(A[k * 14 + i]);
      c[i][0] = _mm512_fmadd_ps(a, b0, c[i][0]);
      c[i][1] = _mm512_fmadd_ps(a, b1, c[i][1]);
    }
  }
  for (int i = 0; i < 14; i++) {
    _mm512_storeu_ps(AB + i * 32, c[i][0]);
    _mm512_storeu_ps(AB + i * 32 + 16, c[i][1]);
  }
}

__attribute__((target("avx2,fma"))) static void uk_f32_fma_6x16(int64_t kc, const float *A,
                                                               const float *B, float *AB) {
  __m256 c[6][2];
  for (int i = 0; i < 6; i++) c[i][0] = c[i][1] = _mm256_setzero_ps();
  for (int64_t k = 0; k < kc; k++) {
    __m256 b0 = _mm256_load_ps(B + k * 16), b1 = _mm256_load_ps(B + k * 16 + 8);
#pragma GCC unroll 6
    for (int i = 0; i < 6; i++) {
      __m256 a = _mm256_set1_ps(A[k * 6 + i]);
      c[i][0] = _mm256_fmadd_ps(a, b0, c[i][0]);
      c[i][1] = _mm256_fmadd_ps(a, b1, c[i][1]);
    }
  }
  for (int i = 0; i < 6; i++) {
    _mm256_storeu_ps(AB + i * 16, c[i][0]);
    _mm256_storeu_ps(AB + i * 16 + 8, c[i][1]);
  }
}

__attribute__((target("avx512f"))) static void uk_f64_avx512_14x16(int64_t kc, const double *A,
                                                                   const double *B, double *AB) {
  __m512d c[14][2];
  for (int i = 0; i < 14; i++) c[i][0] = c[i][1] = _mm512_setzero_pd();
  for (int64_t k = 0; k < kc; k++) {
    __m512d b0 = _mm512_load_pd(B + k * 16), b1 = _mm512_load_pd(B + k * 16 + 8);
#pragma GCC unroll 14
    for (int i = 0; i < 14; i++) {
      __m512d a = _mm512_set1_pd(A[k * 14 + i]);
      c[i][0] = _mm512_fmadd_pd(a, b0, c[i][0]);
      c[i][1] = _mm512_fmadd_pd(a, b1, c[i][1]);
    }
  }
  for (int i = 0; i < 14; i++) {
    _mm512_storeu_pd(AB + i * 16, c[i][0]);
    _mm512_storeu_pd(AB + i * 16 + 8, c[i][1]);
  }
}

__attribute__((target("avx512f"))) static void uk_i32_avx512_14x32(int64_t kc, const int32_t *A,
                                                                   const int32_t *B, int32_t *AB) {
  __m512i c[14][2];
  for (int i = 0; i < 14; i++) c[i][0] = c[i][1] = _mm512_setzero_si512();
  for (int64_t k = 0; k < kc; k++) {
    __m512i b0 = _mm512_load_si512((const void *)(B + k * 32));
    __m512i b1 = _mm512_load_si512((const void *)(B + k * 32 + 16));
#pragma GCC unroll 14
    for (int i = 0; i < 14; i++) {
      __m512i a = _mm512_set1_epi32(A[k * 14 + i]);
      c[i][0] = _mm512_add_epi32(_mm512_mullo_epi32(a, b0), c[i][0]);
      c[i][1] = _mm512_add_epi32(_mm512_mullo_epi32(a, b1), c[i][1]);
    }
  }
  for (int i = 0; i < 14; i++) {
    _mm512_storeu_si512((void *)(AB + i * 32), c[i][0]);
    _mm512_storeu_si512((void *)(AB + i * 32 + 16), c[i][1]);
  }
}

__attribute__((target("avx2"))) static void uk_i32_avx2_6x16(int64_t kc, const int32_t *A,
                                                             const int32_t *B, int32_t *AB) {
  __m256i c[6][2];
  for (int i = 0; i < 6; i++) c[i][0] = c[i][1] = _mm256_setzero_si256();
  for (int64_t k = 0; k < kc; k++) {
    __m256i b0 = _mm256_load_si256((const __m256i *)(B + k * 16));
    __m256i b1 = _mm256_load_si256((const __m256i *)(B + k * 16 + 8));
#pragma GCC unroll 6
    for (int i = 0; i < 6; i++) {
      __m256i a = _mm256_set1_epi32(A[k * 6 + i]);
      c[i][0] = _mm256_add_epi32(_mm256_mullo_epi32(a, b0), c[i][0]);
      c[i][1] = _mm256_add_epi32(_mm256_mullo_epi32(a, b1), c[i][1]);
    }
  }
  for (int i = 0; i < 6; i++) {
    _mm256_storeu_si256((__m256i *)(AB + i * 16), c[i][0]);
    _mm256_storeu_si256((__m256i *)(AB + i * 16 + 8), c[i][1]);
  }
}

static int ukernel_simd_f32(int64_t kc, const float *A, const float *B, float *AB, int MR, int NR,
                            int fused) {
  if (!fused) return 0;
  if (MR == 14 && NR == 32 && has_avx512()) { uk_f32_avx512_14x32(kc, A, B, AB); return 1; }
  if (MR == 6 && NR == 16 && has_fma()) { uk_f32_fma_6x16(kc, A, B, AB); return 1; }
  return 0;
}
static int ukernel_simd_f64(int64_t kc, const double *A, const double *B, double *AB, int MR, int NR,
                            int fused) {
  if (fused && MR == 14 && NR == 16 && has_avx512()) { uk_f64_avx512_14x16(kc, A, B, AB); return 1; }
  return 0;
}
static int ukernel_simd_i32(int64_t kc, const int32_t *A, const int32_t *B, int32_t *AB, int MR,
                            int NR, int fused) {
  (void)fused;
  if (MR == 14 && NR == 32 && has_avx512()) { uk_i32_avx512_14x32(kc, A, B, AB); return 1; }
  if (MR == 6 && NR == 16 && __builtin_cpu_supports("avx2")) { uk_i32_avx2_6x16(kc, A, B, AB); return 1; }
  return 0;
}
static int ukernel_simd_i64(int64_t kc, const int64_t *A, const int64_t *B, int64_t *AB, int MR,
                            int NR, int fused) {
  (void)kc; (void)A; (void)B; (void)AB; (void)MR; (void)NR; (void)fused;
  return 0; /* scalar restatement only */
}

/* ---- instantiate the generic body for the four element types Laser dispatches on
 *      (gemm.nim:228-247: float32, float64, int32, int64) -------------------------------- */

#define T float
#define UT float
#define SFX f32
#define IS_FLOAT 1
#define FMA(a, b, c) fmaf(a, b, c)
#define MULT(a, b) ((a) * (b))
#define ADDT(a, b) ((a) + (b))
#include "laser_gemm_impl.inc"
#undef T
#undef UT
#undef SFX
#undef IS_FLOAT
#undef FMA
#undef MULT
#undef ADDT

#define T double
#define UT double
#define SFX f64
#define IS_FLOAT 1
#define FMA(a, b, c) fma(a, b, c)
#define MULT(a, b) ((a) * (b))
#define ADDT(a, b) ((a) + (b))
#include "laser_gemm_impl.inc"
#undef T
#undef UT
#undef SFX
#undef IS_FLOAT
#undef FMA
#undef MULT
#undef ADDT

#define T int32_t
#define UT uint32_t
#define SFX i32
#define IS_FLOAT 0
#define FMA(a, b, c) 0
#define MULT(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
#define ADDT(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#include "laser_gemm_impl.inc"
#undef T
#undef UT
#undef SFX
#undef IS_FLOAT
#undef FMA
#undef MULT
#undef ADDT

#define T int64_t
#define UT uint64_t
#define SFX i64
#define IS_FLOAT 0
#define FMA(a, b, c) 0
#define MULT(a, b) ((int64_t)((uint64_t)(a) * (uint64_t)(b)))
#define ADDT(a, b) ((int64_t)((uint64_t)(a) + (uint64_t)(b)))
#include "laser_gemm_impl.inc"
#undef T
#undef UT
#undef SFX
#undef IS_FLOAT
#undef FMA
#undef MULT
#undef ADDT

/* ------------------------------------------------------------------------------------------
 * im2col + GEMM convolution -- benchmarks/convolution/conv2d_im2col.nim, conv2d_common.nim
 * ---------------------------------------------------------------------------------------- */

/* conv2d_out_shape -- conv2d_common.nim:15-45 (dilation fixed at 1) */
void oracle_conv2d_out_shape(int64_t iH, int64_t iW, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                             int64_t sH, int64_t sW, int64_t *oH, int64_t *oW) {
  *oH = 1 + (iH + 2 * pH - ((kH - 1) + 1)) / sH;
  *oW = 1 + (iW + 2 * pW - ((kW - 1) + 1)) / sW;
}

/* im2col_workspace_size -- conv2d_im2col.nim:10-20: C * kH * kW * outH * outW (elements) */
int64_t oracle_im2col_workspace_size(int64_t C, int64_t H, int64_t W, int64_t kH, int64_t kW,
                                     int64_t pH, int64_t pW, int64_t sH, int64_t sW) {
  int64_t oH, oW;
  oracle_conv2d_out_shape(H, W, kH, kW, pH, pW, sH, sW, &oH, &oW);
  return C * kH * kW * oH * oW;
}

/* im2col -- conv2d_im2col.nim:42-88: one NCHW image [C,H,W] -> [C*kH*kW, outH*outW], zeros for
 * padding, strides supported, no dilation. */
void oracle_im2col_f32(float *ws, int64_t outH, int64_t outW, const float *input, int64_t C,
                       int64_t H, int64_t W, int64_t kH, int64_t kW, int64_t pH, int64_t pW,
                       int64_t sH, int64_t sW) {
  const float *in = input;
  for (int64_t c = 0; c < C; c++) {
    for (int64_t krow = 0; krow < kH; krow++)
      for (int64_t kcol = 0; kcol < kW; kcol++) {
        int64_t row = -pH + krow;
        for (int64_t oh = 0; oh < outH; oh++) {
          if (!(row >= 0 && row < H)) {
            for (int64_t ow = 0; ow < outW; ow++) *ws++ = 0.0f;
          } else {
            int64_t col = -pW + kcol;
            for (int64_t ow = 0; ow < outW; ow++) {
              *ws++ = (col >= 0 && col < W) ? in[row * W + col] : 0.0f;
              col += sW;
            }
          }
          row += sH;
        }
      }
    in += H * W;
  }
}

/* conv2d_im2col -- conv2d_im2col.nim:90-166: per image n: im2col (skipped for 1x1) then
 * out[n] (C_out x oH*oW) = kernel (C_out x C_in*kH*kW) . workspace, alpha = 1, beta = 0.
 * The reference makes that call through OpenBLAS cblas_sgemm (benchmarks/third_party/blas.nim:
 * 18-20, un-pinned third-party); here it goes through the Laser GEMM restated above, which is
 * the wiring the north star asks for.  Integer-valued KATs (conv2d_common.nim:147-283) are
 * exact either way. */
int oracle_conv2d_im2col_f32(float *output, const float *input, int64_t Nb, int64_t C_in, int64_t H,
                             int64_t W, const float *kernel, int64_t C_out, int64_t kH, int64_t kW,
                             int64_t pH, int64_t pW, int64_t sH, int64_t sW, float *workspace,
                             int isa) {
  int64_t oH, oW;
  oracle_conv2d_out_shape(H, W, kH, kW, pH, pW, sH, sW, &oH, &oW);
  int is1x1 = (kH * kW == 1);
  for (int64_t n = 0; n < Nb; n++) {
    const float *pin = input + n * C_in * H * W;
    if (!is1x1) oracle_im2col_f32(workspace, oH, oW, pin, C_in, H, W, kH, kW, pH, pW, sH, sW);
    const float *ws = is1x1 ? pin : workspace;
    float *pout = output + n * C_out * oH * oW;
    int64_t M = C_out, K = C_in * kH * kW, N = oH * oW;
    int rc = oracle_gemm_strided_f32(M, N, K, 1.0f, kernel, K, 1, ws, N, 1, 0.0f, pout, N, 1, isa, 1);
    if (rc) return rc;
  }
  return 0;
}

/* Independent cross-check: direct convolution -- benchmarks/convolution/
 * conv2d_direct_convolution.nim:50-73 (we use sW for the column stride; the reference uses sH
 * there, which only matters when sH != sW).  float64 accumulation. */
void oracle_conv2d_direct_f32(float *output, const float *input, int64_t Nb, int64_t C_in, int64_t H,
                              int64_t W, const float *kernel, int64_t C_out, int64_t kH, int64_t kW,
                              int64_t pH, int64_t pW, int64_t sH, int64_t sW) {
  int64_t oH, oW;
  oracle_conv2d_out_shape(H, W, kH, kW, pH, pW, sH, sW, &oH, &oW);
#pragma omp parallel for collapse(2)
  for (int64_t n = 0; n < Nb; n++)
    for (int64_t co = 0; co < C_out; co++)
      for (int64_t oh = 0; oh < oH; oh++)
        for (int64_t ow = 0; ow < oW; ow++) {
          double acc = 0.0;
          for (int64_t ci = 0; ci < C_in; ci++)
            for (int64_t kh = 0; kh < kH; kh++)
              for (int64_t kw = 0; kw < kW; kw++) {
                int64_t r = oh * sH - pH + kh, c = ow * sW - pW + kw;
                if (r >= 0 && r < H && c >= 0 && c < W)
                  acc += (double)input[((n * C_in + ci) * H + r) * W + c] *
                         (double)kernel[((co * C_in + ci) * kH + kh) * kW + kw];
              }
          output[((n * C_out + co) * oH + oh) * oW + ow] = (float)acc;
        }
}

/* mean_relative_error -- laser/private/error_functions.nim:6-26 (accumulated in the element
 * type, like the reference) */
float oracle_mean_relative_error_f32(const float *y, const float *y_true, int64_t n) {
  float r = 0.0f;
  for (int64_t i = 0; i < n; i++) {
    float a = fabsf(y_true[i]), b = fabsf(y[i]);
    float denom = a > b ? a : b;
    if (denom != 0.0f) r += fabsf(y_true[i] - y[i]) / denom;
  }
  return r / (float)n;
}

/* float64 naive product ("Reference loop", benchmarks/gemm/gemm_bench_float32.nim:116-137):
 * an independent cross-check for the restatement itself. */
void oracle_naive_gemm_f64acc(int64_t M, int64_t N, int64_t K, const float *A, int64_t rsA,
                              int64_t csA, const float *B, int64_t rsB, int64_t csB, double *C) {
#pragma omp parallel for
  for (int64_t i = 0; i < M; i++)
    for (int64_t j = 0; j < N; j++) {
      double acc = 0.0;
      for (int64_t k = 0; k < K; k++) acc += (double)A[i * rsA + k * csA] * (double)B[k * rsB + j * csB];
      C[i * N + j] = acc;
    }
}
