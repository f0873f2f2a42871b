// Micro-probe for v_mfma_f32_4x4x1_16B_f32 on gfx950 (round 4): operand / result lane layout, the CBSZ / ABID broadcast of the A operand, and the
// issue rate with 1..8 independent accumulators, with LDS operand reads and with a global filter stream in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma4x4.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- layout: D = A (x) B with one-hot lanes.  out[(la * 64 + lb) * 256 + r * 64 + lane] ------------------------------------------------------------
template <int CBSZ, int ABID>
__global__ void layout_kernel(float* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const float a = lane == la ? 1.f : 0.f, b = lane == lb ? 1.f : 0.f;
      f4 d = {0.f, 0.f, 0.f, 0.f};
      d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, CBSZ, ABID, 0);
      for (int r = 0; r < 4; ++r) out[((size_t)(la * 64 + lb) * 4 + r) * 64 + lane] = d[r];
    }
}

template <int CBSZ, int ABID>
void layout(const char* name) {
  float* out;
  const size_t n = (size_t)64 * 64 * 256;
  (void)hipMalloc(&out, n * sizeof(float));
  layout_kernel<CBSZ, ABID><<<1, 64>>>(out);
  std::vector<float> h(n);
  (void)hipMemcpy(h.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  // hypothesis: D[r][lane] += A[(CBSZ ? ABID : lane / 4) * 4 + r] * B[lane]   (block = lane / 4, row i = VGPR r, column j = lane % 4)
  long bad = 0, nz = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int r = 0; r < 4; ++r)
        for (int lane = 0; lane < 64; ++lane) {
          const float v = h[((size_t)(la * 64 + lb) * 4 + r) * 64 + lane];
          const int ablk = CBSZ == 4 ? ABID : lane / 4;
          const float e = (lb == lane && la == ablk * 4 + r) ? 1.f : 0.f;
          nz += v != 0.f;
          bad += v != e;
        }
  printf("layout %-24s nonzeros %ld  mismatches vs hypothesis %ld\n", name, nz, bad);
  if (bad) {   // dump the true map compactly: for each (la, lb) the (r, lane) that lit up
    int shown = 0;
    for (int la = 0; la < 64 && shown < 48; ++la)
      for (int lb = 0; lb < 64 && shown < 48; ++lb)
        for (int r = 0; r < 4; ++r)
          for (int lane = 0; lane < 64; ++lane)
            if (h[((size_t)(la * 64 + lb) * 4 + r) * 64 + lane] != 0.f && shown < 48) {
              printf("   A lane %2d x B lane %2d -> D[vgpr %d][lane %2d]\n", la, lb, r, lane);
              ++shown;
            }
  }
  (void)hipFree(out);
}

// ---- rate ---------------------------------------------------------------------------------------------------------------------------------------------
// MODE 0: MFMA only, NACC independent accumulators, A broadcast by abid 0..15 (16 MFMAs per A register)
// MODE 1: + one ds_read_b128 (the pixel operand: 4 channels) per 4 * NACC MFMAs
// MODE 2: + A registers streamed from global memory (one dwordx4 per 4 accumulators x 16 MFMAs), prefetched one step ahead
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, const float* __restrict__ wstream) {
  __shared__ __attribute__((aligned(16))) float lds[12 * 1024];
  f4 acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
  for (int e = threadIdx.x; e < 12 * 1024; e += 256) lds[e] = (float)(e & 15) * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a[NACC];
  for (int q = 0; q < NACC; ++q) a[q] = 0.5f + 0.001f * (float)(lane + q);
  f4 an[(NACC + 3) / 4];
  const f4* ws = reinterpret_cast<const f4*>(wstream) + lane;
  f4 b = {1.f, 0.5f, 0.25f, 0.125f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int v = 0; v < (NACC + 3) / 4; ++v) an[v] = ws[((it * ((NACC + 3) / 4) + v) & 1023) * 64];
    }
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) {      // 4 channel quads of a 16-channel step: 16 abid values
      if (MODE >= 1) b = *reinterpret_cast<const f4*>(&lds[((lane * 20 + cq * 4 + it * 80) % (12 * 1024 - 4)) & ~3]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
          switch (cq * 4 + c) {
#define CASE(K) case K: acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q], b[c], acc[q], 4, K, 0); break;
            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
          }
        }
    }
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < NACC; ++q) a[q] = an[q / 4][q & 3];
    }
  }
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < NACC; ++q) s += acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, int MODE>
void rate(const char* name, int blocks_per_cu) {
  float *out, *ws;
  const int grid = 256 * blocks_per_cu, iters = 4000;
  (void)hipMalloc(&out, grid * 256 * sizeof(float));
  (void)hipMalloc(&ws, 1024 * 64 * 16);
  (void)hipMemset(ws, 0, 1024 * 64 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  rate_kernel<NACC, MODE><<<grid, 256>>>(out, 10, ws);
  (void)hipEventRecord(e0);
  rate_kernel<NACC, MODE><<<grid, 256>>>(out, iters, ws);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * iters * 16.0 * NACC;
  printf("rate %-30s acc=%d blocks/CU=%d  %8.3f ms  %7.1f TFLOP/s  %5.2f clk/MFMA/SIMD @2.4GHz\n", name, NACC, blocks_per_cu, ms, mfma * 512.0 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
  (void)hipFree(out);
  (void)hipFree(ws);
}

int main() {
  layout<0, 0>("cbsz=0");
  layout<4, 0>("cbsz=4 abid=0");
  layout<4, 5>("cbsz=4 abid=5");
  layout<4, 15>("cbsz=4 abid=15");
  for (int b = 1; b <= 2; ++b) {
    rate<1, 0>("mfma only", b);
    rate<2, 0>("mfma only", b);
    rate<4, 0>("mfma only", b);
    rate<6, 0>("mfma only", b);
    rate<10, 0>("mfma only", b);
    rate<20, 0>("mfma only", b);
    rate<1, 1>("+ ds_read_b128 / 4 mfma", b);
    rate<4, 1>("+ ds_read_b128 / 16 mfma", b);
    rate<6, 1>("+ ds_read_b128 / 24 mfma", b);
    rate<10, 1>("+ ds_read_b128 / 40 mfma", b);
    rate<20, 1>("+ ds_read_b128 / 80 mfma", b);
    rate<6, 2>("+ ds_read + A stream", b);
    rate<10, 2>("+ ds_read + A stream", b);
    rate<20, 2>("+ ds_read + A stream", b);
  }
  return 0;
}
