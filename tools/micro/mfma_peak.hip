// Microbenchmark: what does the 16x16x4 fp32 MFMA stream of the conv kernels reach on this chip?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: MFMA only; 1: + 8 ds_read_b128 per 64 MFMA; 2: + barrier; 3: + ~200 VALU int ops; 4: 2 + 4 global float4 loads; 5: 4 + 4 ds_write_b128 (double buffered); 6: 5 + 25 VALU
__global__ __launch_bounds__(256) void k(float* out, int iters, int salt, const float* src) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  f4 fa[4], fb[4];
  for (int i = 0; i < 4; ++i) {
    fa[i] = f4{1.f, 2.f, 3.f, 4.f} * (float)(threadIdx.x + i);
    fb[i] = f4{4.f, 3.f, 2.f, 1.f} * (float)(threadIdx.x + i + salt);
  }
  for (int e = threadIdx.x; e < 8192; e += 256) lds[e] = (float)e;
  __syncthreads();
  int junk = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
      if (MODE == 7) {   // the conv kernel's fragment addressing: 128-row x 16-float tiles, XOR-swizzled quads
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
        const float* A = lds + (it & 1) * 4096;
        const float* B = A + 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (wave >> 1) * 64 + i * 16 + lr;
          fa[i] = *reinterpret_cast<const f4*>(A + (r * 4 + (lq ^ ((r >> 1) & 3))) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (wave & 1) * 64 + i * 16 + lr;
          fb[i] = *reinterpret_cast<const f4*>(B + (r * 4 + (lq ^ ((r >> 1) & 3))) * 4);
        }
      } else {
      const int base = ((threadIdx.x & 63) * 4 + it * 4) & 2047;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f4*>(&lds[(base + i * 1024) & 8188]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const f4*>(&lds[(base + 512 + i * 1024) & 8188]);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int v = 0; v < 200; ++v) junk = junk * 1664525 + (junk >> 3) + v;
    }
    f4 g[4];
    if (MODE >= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = *reinterpret_cast<const f4*>(src + ((size_t)(blockIdx.x * 131 + it * 17 + i * 5) % 4096) * 1024 + threadIdx.x * 4);
    }
    if (MODE == 6) {
#pragma unroll
      for (int v = 0; v < 25; ++v) junk = junk + (junk >> 3) + v;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    if (MODE >= 5) {
      const int wb = ((it & 1) ? 4096 : 0) + threadIdx.x * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f4*>(&lds[(wb + i * 1024) & 8188]) = g[i];
    } else if (MODE == 4) {
      asm volatile("" ::"v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
    }
    if (MODE >= 2) __syncthreads();
  }
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (float)junk;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  float* out;
  float* src;
  (void)hipMalloc(&src, 4096 * 1024 * sizeof(float));
  (void)hipMemset(src, 0, 4096 * 1024 * sizeof(float));
  const int grid = 256 * blocks_per_cu, iters = 4000;
  (void)hipMalloc(&out, grid * 256 * sizeof(float));
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  k<MODE><<<grid, 256>>>(out, 10, 1, src);
  (void)hipEventRecord(a);
  k<MODE><<<grid, 256>>>(out, iters, 1, src);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 4 /*waves*/ * iters * 64 /*mfma*/ * 2048.0;
  printf("%-34s blocks/CU=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  (void)hipFree(out);
  (void)hipFree(src);
}

int main() {
  for (int b = 2; b <= 3; ++b) {
    run<0>("mfma only", b);
    run<2>("mfma + ds_read + barrier", b);
    run<4>("  + 4 global_load_dwordx4", b);
    run<5>("  + 4 ds_write_b128", b);
    run<6>("  + 25 VALU", b);
    run<7>("mfma + SWIZZLED ds_read + barrier", b);
  }
  return 0;
}
