// Micro-probe for split-bf16 ("bf16x3" / "bf16x6") fp32 GEMM tiles on gfx950 (round 4, exploratory -- DESIGN section 6): does the matrix pipe
// deliver the arithmetic tools/bf16x3_numerics.py restates on the host, and at what rate?
//   1. tile: C[32][32] = A[32][K] * B[K][32] from fp32 operands split on the device (x = a1 + a2 (+ a3), ai = bf16 of the running residual, round to
//      nearest even), 3 or 6 products per 16-deep slice on v_mfma_f32_32x32x16_bf16, against an fp64 product on the host: checks the operand /
//      result lane layout (asymmetric operands) and the error classes (2^-16 / 2^-24 of sum |a b|);
//   2. rate: register-resident streams of 3 and 6 products per operand slice with 4 independent accumulators, with and without the split
//      (12 VALU per element pair) inside the loop, in fp32-EQUIVALENT TFLOP/s (one fp32 multiply-add = 3 or 6 bf16 ones) beside the chip's
//      fp32 MFMA peak of 157.3.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16x3.hip -o /tmp/mfma_bf16x3 && /tmp/mfma_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

union Frag {
  bf8 v;
  unsigned short h[8];
};

// x[8] -> PARTS bf16 fragments
template <int PARTS>
__device__ __forceinline__ void split8(const float (&x)[8], Frag (&out)[PARTS]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = x[i];
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
      const unsigned short h = bf16_rne(r);
      out[p].h[i] = h;
      r -= bf16_f32(h);
    }
  }
}

// ---- 1. one 32 x 32 tile ----------------------------------------------------------------------------------------------------------------------------------
// assumed layout of v_mfma_f32_32x32x16_bf16: A fragment = A[row lane & 31][k (lane >> 5) * 8 + i], B fragment = B[k (lane >> 5) * 8 + i][col lane & 31],
// D[reg] = C[row (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)][col lane & 31]
template <int TERMS>
__global__ void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
  const int lane = threadIdx.x, rc = lane & 31, kg = lane >> 5;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[i] = A[rc * K + k0 + kg * 8 + i];
      b[i] = B[(k0 + kg * 8 + i) * 32 + rc];
    }
    constexpr int PARTS = TERMS == 6 ? 3 : (TERMS == 1 ? 1 : 2);
    Frag fa[PARTS], fb[PARTS];
    split8<PARTS>(a, fa);
    split8<PARTS>(b, fb);
    // smallest products first
    if constexpr (TERMS == 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2].v, fb[0].v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[2].v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, fb[1].v, acc, 0, 0, 0);
    }
    if constexpr (TERMS >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, fb[0].v, acc, 0, 0, 0);
    if constexpr (TERMS >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[0].v, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + rc] = acc[r];
}

template <int TERMS>
void tile(int K, const std::vector<float>& hA, const std::vector<float>& hB) {
  float *A, *B, *C;
  (void)hipMalloc(&A, hA.size() * 4);
  (void)hipMalloc(&B, hB.size() * 4);
  (void)hipMalloc(&C, 32 * 32 * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  tile_kernel<TERMS><<<1, 64>>>(A, B, C, K);
  std::vector<float> hC(32 * 32);
  (void)hipMemcpy(hC.data(), C, 32 * 32 * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst32 = 0, scale = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0, sa = 0;
      float s32 = 0.f;
      for (int k = 0; k < K; ++k) {
        s += (double)hA[i * K + k] * hB[k * 32 + j];
        sa += fabs((double)hA[i * K + k] * hB[k * 32 + j]);
        s32 = fmaf(hA[i * K + k], hB[k * 32 + j], s32);
      }
      worst = fmax(worst, fabs(hC[i * 32 + j] - s) / sa);
      worst32 = fmax(worst32, fabs((double)s32 - s) / sa);
      scale = fmax(scale, sa);
    }
  printf("tile K=%4d  %d products: worst |C - exact| / sum|a b| = %.3e  (2^%.1f)   [host fp32 fma chain: %.3e]\n", K, TERMS, worst, log2(worst), worst32);
  (void)hipFree(A);
  (void)hipFree(B);
  (void)hipFree(C);
}

// ---- 2. rate ---------------------------------------------------------------------------------------------------------------------------------------------
// TERMS products per operand slice on NACC independent 32 x 32 accumulators; SPLIT: the fp32 -> bf16 parts conversion of one A and one B slice
// (8 + 8 floats per lane) inside the loop, as a kernel that splits while staging would pay it
template <int TERMS, int SPLIT>
__global__ __launch_bounds__(256) void rate_kernel(const float* __restrict__ src, float* __restrict__ out, int iters) {
  constexpr int NACC = 4, PARTS = TERMS == 6 ? 3 : 2;
  f16v acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[threadIdx.x * 8 + i];
    b[i] = src[2048 + threadIdx.x * 8 + i];
  }
  Frag fa[PARTS], fb[PARTS];
  split8<PARTS>(a, fa);
  split8<PARTS>(b, fb);
  for (int it = 0; it < iters; ++it) {
    if (SPLIT) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[i] += 1.0f;      // keeps the conversion inside the loop
        b[i] -= 1.0f;
      }
      split8<PARTS>(a, fa);
      split8<PARTS>(b, fb);
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      if constexpr (TERMS == 6) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2].v, fb[0].v, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[2].v, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, fb[1].v, acc[q], 0, 0, 0);
      }
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1].v, fb[0].v, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[1].v, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0].v, fb[0].v, acc[q], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TERMS, int SPLIT>
void rate(const float* src, float* out, int wg_per_cu) {
  const int iters = 4096, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  rate_kernel<TERMS, SPLIT><<<grid, 256>>>(src, out, 64);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  rate_kernel<TERMS, SPLIT><<<grid, 256>>>(src, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * iters * 4 * TERMS;       // waves x iterations x accumulators x products
  const double flop_bf16 = mfma * 2.0 * 32 * 32 * 16;
  printf("rate  %d products%s, %d WG/CU: %7.1f bf16 TFLOP/s = %6.1f fp32-equivalent TFLOP/s (fp32 MFMA peak 157.3)   %.2f ms\n", TERMS,
         SPLIT ? " + split in the loop" : "                    ", wg_per_cu, flop_bf16 / ms * 1e-9, flop_bf16 / TERMS / ms * 1e-9, ms);
}

int main() {
  srand(7);
  for (int K : {16, 512, 2048, 8192}) {
    std::vector<float> hA(32 * K), hB(K * 32);
    for (auto& v : hA) v = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 0.7f;       // asymmetric, non-zero mean
    for (auto& v : hB) v = (float)(rand() & 0xffffff) / 16777216.f * 3.f - 1.1f;
    tile<1>(K, hA, hB);
    tile<3>(K, hA, hB);
    tile<6>(K, hA, hB);
  }
  float *src, *out;
  (void)hipMalloc(&src, 4096 * 4);
  (void)hipMalloc(&out, 256 * 2 * 256 * 4);
  std::vector<float> h(4096);
  for (auto& v : h) v = (float)(rand() & 0xffffff) / 16777216.f;
  (void)hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    rate<3, 0>(src, out, w);
    rate<6, 0>(src, out, w);
    rate<3, 1>(src, out, w);
    rate<6, 1>(src, out, w);
  }
  return 0;
}
