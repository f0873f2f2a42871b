// Prototype (round 4, exploratory -- DESIGN section 6): an LDS-fed fp32-class GEMM tile on the bf16 matrix pipe, C[M][N] = A[M][K] * B[N][K]^T with
// PRE-SPLIT operands (x = x1 + x2 (+ x3), bf16 planes), 3 or 6 products per 16-deep slice on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// Deliberately the simple structure of the library's wide fp32 tiles (128 x 128 workgroup tile, 2 x 2 waves of 64 x 64, BK = 64, two barriers per
// chunk, register-staged double buffer) so that the number it prints is what a first bf16x3 conv_dgrad32d / conv_wgrad32d could expect, in
// fp32-EQUIVALENT TFLOP/s against the 126-134 the fp32 tiles reach today (peak 157.3).  Checks itself against an fp64 product on a small case.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_bf16x3.hip -o /tmp/gemm_bf16x3 && /tmp/gemm_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int PLANE = 128 * BK * 2;              // bytes of one 128-row operand plane of a chunk (16 KB)

// PARTS planes per operand: A planes [p][M][K], B planes [p][N][K] (bf16).  LDS: [buffer][A planes | B planes][128 rows][8 chunks of 16 B], the chunk
// index XOR-ed with (row & 7) (a 32-row fragment read of one logical chunk then covers all eight 16-B slots of the 128-B bank row).
template <int PARTS>
__global__ __launch_bounds__(256) void gemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NPL = 2 * PARTS, BUF = NPL * PLANE, LD = NPL * 4;     // 16-B loads per thread and chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  // staging map: thread -> chunk c = tid & 7 of rows (tid >> 3) + 32 j
  const int sc = tid & 7, sr = tid >> 3;
  u4 stg[LD];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const bf16_t* base = p < PARTS ? A + ((size_t)p * M + m0) * K : B + ((size_t)(p - PARTS) * N + n0) * K;
#pragma unroll
      for (int j = 0; j < 4; ++j) stg[p * 4 + j] = *reinterpret_cast<const u4*>(base + (size_t)(sr + 32 * j) * K + k0 + sc * 8);
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* d = smem + buf * BUF;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = sr + 32 * j;
        *reinterpret_cast<u4*>(d + p * PLANE + r * 128 + ((sc ^ (r & 7)) << 4)) = stg[p * 4 + j];
      }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, kg = lane >> 5;
  auto mma = [&](int buf) {
    const unsigned char* s = smem + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf8 fa[PARTS][2], fb[PARTS][2];
      const int c = ks * 2 + kg;
#pragma unroll
      for (int p = 0; p < PARTS; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
          fa[p][i] = *reinterpret_cast<const bf8*>(s + p * PLANE + ra * 128 + ((c ^ (ra & 7)) << 4));
          fb[p][i] = *reinterpret_cast<const bf8*>(s + (PARTS + p) * PLANE + rb * 128 + ((c ^ (rb & 7)) << 4));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (PARTS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const int nk = K / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    // two LDS buffers with two planes per operand (128 KB); one with three (96 KB: the next chunk waits in registers for a second barrier)
    const int buf = PARTS == 2 ? (kc & 1) : 0;
    if (kc + 1 < nk) gload((kc + 1) * BK);      // in flight behind this chunk's MFMA stream
    mma(buf);
    if (PARTS == 3) __syncthreads();
    if (kc + 1 < nk) sstore(PARTS == 2 ? buf ^ 1 : 0);      // two buffers: buf ^ 1 was last read before the previous barrier
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, col = n0 + wn * 64 + j * 32 + fr;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// The same tile with the operand planes DMA-ed straight into LDS (buffer_load_dwordx4 ... lds, as conv_fwd32d / dgrad32d / wgrad32d stage their fp32 tiles):
// one instruction writes 8 rows x 128 B lane-linearly, so the XOR swizzle is applied to the SOURCE chunk a lane fetches; no staging registers, no
// ds_write pass, ONE barrier per chunk (issue next chunk -> MFMA stream of this one -> vmcnt(0) -> barrier).
template <int PARTS>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  constexpr int NPL = 2 * PARTS, BUF = NPL * PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, 0x7fffffff, 0x00020000);
  // wave w, instruction i -> rows (w * 4 + i) * 8 + (lane >> 3) of a plane; LDS slot lane & 7 <- source chunk slot ^ (row & 7)
  unsigned voA[4], voB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const unsigned c = (unsigned)((lane & 7) ^ (row & 7));
    voA[i] = ((unsigned)(m0 + row) * (unsigned)K + c * 8u) * 2u;
    voB[i] = ((unsigned)(n0 + row) * (unsigned)K + c * 8u) * 2u;
  }
  const unsigned plA = (unsigned)M * (unsigned)K * 2u, plB = (unsigned)N * (unsigned)K * 2u;     // bytes between planes
  auto issue = [&](int buf, int k0) {
    unsigned char* d = smem + buf * BUF + wave * 4 * 1024;
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_t)(d + p * PLANE + i * 1024), 16, voA[i], (unsigned)p * plA + (unsigned)k0 * 2u, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_t)(d + (PARTS + p) * PLANE + i * 1024), 16, voB[i], (unsigned)p * plB + (unsigned)k0 * 2u, 0, 0);
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, kg = lane >> 5;
  auto mma = [&](int buf) {
    const unsigned char* s = smem + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf8 fa[PARTS][2], fb[PARTS][2];
      const int c = ks * 2 + kg;
#pragma unroll
      for (int p = 0; p < PARTS; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
          fa[p][i] = *reinterpret_cast<const bf8*>(s + p * PLANE + ra * 128 + ((c ^ (ra & 7)) << 4));
          fb[p][i] = *reinterpret_cast<const bf8*>(s + (PARTS + p) * PLANE + rb * 128 + ((c ^ (rb & 7)) << 4));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const int nk = K / BK;
  issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kc = 0; kc + 1 < nk; ++kc) {
    const int buf = kc & 1;
    issue(buf ^ 1, (kc + 1) * BK);      // buf ^ 1 was released by the barrier below
    mma(buf);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  mma((nk - 1) & 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, col = n0 + wn * 64 + j * 32 + fr;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// The direct-to-LDS tile with 32-deep chunks: 64-byte rows, 64 KB of LDS per workgroup -> TWO workgroups per CU (two waves per SIMD, as the library's fp32
// tiles run): one workgroup's barrier / DMA wait is the other's MFMA stream.  Swizzle for 64-byte rows: 16-byte piece c of row r sits at slot c ^ ((r >> 2) & 3)
// (rows r, r + 4, r + 8, r + 12 share a 64-byte quarter of the 256-byte bank row).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dma32_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                                                                    float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  constexpr int PL = 128 * 64, BUF = 4 * PL;      // plane = 128 rows x 64 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, 0x7fffffff, 0x00020000);
  // wave w, instruction i (2 per plane) -> rows (w * 2 + i) * 16 + (lane >> 2); LDS slot lane & 3 <- source piece slot ^ ((row >> 2) & 3) = slot ^ ((lane >> 4) & 3)
  unsigned voA[2], voB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const unsigned c = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    voA[i] = ((unsigned)(m0 + row) * (unsigned)K + c * 8u) * 2u;
    voB[i] = ((unsigned)(n0 + row) * (unsigned)K + c * 8u) * 2u;
  }
  const unsigned plA = (unsigned)M * (unsigned)K * 2u, plB = (unsigned)N * (unsigned)K * 2u;
  auto issue = [&](int buf, int k0) {
    unsigned char* d = smem + buf * BUF + wave * 2 * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_t)(d + p * PL + i * 1024), 16, voA[i], (unsigned)p * plA + (unsigned)k0 * 2u, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_t)(d + (2 + p) * PL + i * 1024), 16, voB[i], (unsigned)p * plB + (unsigned)k0 * 2u, 0, 0);
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, kg = lane >> 5;
  auto mma = [&](int buf) {
    const unsigned char* s = smem + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8 fa[2][2], fb[2][2];
      const int c = ks * 2 + kg;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
          fa[p][i] = *reinterpret_cast<const bf8*>(s + p * PL + ra * 64 + ((c ^ ((ra >> 2) & 3)) << 4));
          fb[p][i] = *reinterpret_cast<const bf8*>(s + (2 + p) * PL + rb * 64 + ((c ^ ((rb >> 2) & 3)) << 4));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const int nk = K / 32;
  issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kc = 0; kc + 1 < nk; ++kc) {
    const int buf = kc & 1;
    issue(buf ^ 1, (kc + 1) * 32);
    mma(buf);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  mma((nk - 1) & 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, col = n0 + wn * 64 + j * 32 + fr;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

static bf16_t bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static float bf16_f32(bf16_t h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static void split_planes(const std::vector<float>& x, int parts, std::vector<bf16_t>& out) {
  out.resize((size_t)parts * x.size());
  for (size_t i = 0; i < x.size(); ++i) {
    float r = x[i];
    for (int p = 0; p < parts; ++p) {
      const bf16_t h = bf16_rne(r);
      out[(size_t)p * x.size() + i] = h;
      r -= bf16_f32(h);
    }
  }
}

__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed) {      // random-like bf16 bit patterns in [-2, 2): timing runs only
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = (bf16_t)((h & 0x80ffu) | (((h >> 16) & 1u ? 0x3f00u : 0x3f80u)));
  }
}

// fp32 tensor -> PARTS bf16 planes (round to nearest even of the running residual): the pass a producer-side epilogue would otherwise do
template <int PARTS>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, size_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f4 r = reinterpret_cast<const f4*>(x)[i];
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
      us4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned u = __float_as_uint(r[e]);
        u += 0x7fffu + ((u >> 16) & 1u);
        h[e] = (unsigned short)(u >> 16);
        r[e] -= __uint_as_float((unsigned)h[e] << 16);
      }
      reinterpret_cast<us4*>(out + (size_t)p * n4 * 4)[i] = h;
    }
  }
}

template <int PARTS>
static void time_split(size_t n) {
  float* x;
  bf16_t* o;
  (void)hipMalloc(&x, n * 4);
  (void)hipMalloc(&o, n * 2 * PARTS);
  std::vector<float> h(1 << 20);
  for (auto& v : h) v = (float)(rand() & 0xffffff) / 16777216.f * 4.f - 1.3f;
  for (size_t off = 0; off < n; off += h.size()) (void)hipMemcpy(x + off, h.data(), (n - off < h.size() ? n - off : h.size()) * 4, hipMemcpyHostToDevice);
  split_kernel<PARTS><<<4096, 256>>>(x, o, n / 4);
  (void)hipDeviceSynchronize();
  std::vector<bf16_t> dev((size_t)PARTS * h.size()), ref;
  for (int p = 0; p < PARTS; ++p) (void)hipMemcpy(dev.data() + (size_t)p * h.size(), o + (size_t)p * n, h.size() * 2, hipMemcpyDeviceToHost);
  split_planes(h, PARTS, ref);
  size_t bad = 0;
  for (size_t i = 0; i < dev.size(); ++i) bad += dev[i] != ref[i];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) split_kernel<PARTS><<<4096, 256>>>(x, o, n / 4);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("split %zu MB fp32 -> %d bf16 planes: %6.1f us  %5.2f TB/s (read + written)   device vs host split: %zu of %zu values differ\n", n * 4 >> 20, PARTS, ms * 1e3,
         (double)n * (4 + 2 * PARTS) / ms * 1e-9, bad, dev.size());
  (void)hipFree(x);
  (void)hipFree(o);
}

static int g_dma = 0;      // 1: the direct-to-LDS variant (two planes per operand only); 2: + 32-deep chunks, two workgroups per CU
template <int PARTS>
static void launch(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K) {
  const int lds = (PARTS == 2 ? 2 : 1) * 2 * PARTS * PLANE, grid = (M / BM) * (N / BN);
  if (g_dma == 2 && PARTS == 2) {
    (void)hipFuncSetAttribute((const void*)gemm_dma32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    gemm_dma32_kernel<<<grid, 256, 64 * 1024>>>(A, B, C, M, N, K);
    return;
  }
  if (g_dma && PARTS == 2) {
    (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    gemm_dma_kernel<2><<<grid, 256, lds>>>(A, B, C, M, N, K);
    return;
  }
  (void)hipFuncSetAttribute((const void*)gemm_kernel<PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  gemm_kernel<PARTS><<<grid, 256, lds>>>(A, B, C, M, N, K);
}

template <int PARTS>
static void check(int M, int N, int K) {
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  for (auto& v : hA) v = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 0.7f;      // asymmetric operands, non-zero means
  for (auto& v : hB) v = (float)(rand() & 0xffffff) / 16777216.f * 3.f - 1.1f;
  std::vector<bf16_t> pA, pB;
  split_planes(hA, PARTS, pA);
  split_planes(hB, PARTS, pB);
  bf16_t *A, *B;
  float* C;
  (void)hipMalloc(&A, pA.size() * 2);
  (void)hipMalloc(&B, pB.size() * 2);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemcpy(A, pA.data(), pA.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, pB.data(), pB.size() * 2, hipMemcpyHostToDevice);
  launch<PARTS>(A, B, C, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  std::vector<float> hC((size_t)M * N);
  (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst32 = 0;
  for (int i = 0; i < M; i += 3)
    for (int j = 0; j < N; j += 5) {
      double s = 0, sa = 0;
      float s32 = 0.f;
      for (int k = 0; k < K; ++k) {
        const double t = (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        s += t;
        sa += fabs(t);
        s32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], s32);
      }
      worst = fmax(worst, fabs(hC[(size_t)i * N + j] - s) / sa);
      worst32 = fmax(worst32, fabs((double)s32 - s) / sa);
    }
  printf("check%s %d x %d x %d, %d products: worst |C - exact| / sum|a b| = %.3e  [host fp32 fma chain %.3e]\n", g_dma && PARTS == 2 ? (g_dma == 2 ? " (direct-to-LDS, BK 32, 2 WG/CU)" : " (direct-to-LDS)") : "", M, N, K, PARTS == 3 ? 6 : 3, worst, worst32);
  (void)hipFree(A);
  (void)hipFree(B);
  (void)hipFree(C);
}

template <int PARTS>
static void timeit(int M, int N, int K) {
  bf16_t *A, *B;
  float* C;
  const size_t na = (size_t)PARTS * M * K, nb = (size_t)PARTS * N * K;
  (void)hipMalloc(&A, na * 2);
  (void)hipMalloc(&B, nb * 2);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  fill_kernel<<<2048, 256>>>(A, na, 1u);
  fill_kernel<<<2048, 256>>>(B, nb, 77u);
  launch<PARTS>(A, B, C, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 5;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch<PARTS>(A, B, C, M, N, K);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * M * N * K;
  printf("time%s  %6d x %5d x %5d, %d products, %5d workgroups: %8.1f us  %6.1f fp32-equivalent TFLOP/s  (%7.1f bf16 TFLOP/s)\n", g_dma && PARTS == 2 ? (g_dma == 2 ? " (direct-to-LDS, BK 32, 2 WG/CU)" : " (direct-to-LDS)") : "", M, N, K, PARTS == 3 ? 6 : 3,
         (M / BM) * (N / BN), ms * 1e3, flop / ms * 1e-9, flop * (PARTS == 3 ? 6 : 3) / ms * 1e-9);
  (void)hipFree(A);
  (void)hipFree(B);
  (void)hipFree(C);
}

int main() {
  srand(11);
  check<2>(256, 384, 512);
  check<3>(256, 384, 512);
  // the PatchGAN layers' GEMM shapes at batch 16 (pixels x Cout x taps * Cin) and a square case
  timeit<2>(65536, 256, 2048);
  timeit<2>(16384, 512, 4096);
  timeit<2>(15360, 512, 8192);
  timeit<2>(8192, 8192, 4096);
  timeit<3>(65536, 256, 2048);
  timeit<3>(15360, 512, 8192);
  timeit<3>(8192, 8192, 4096);
  g_dma = 1;
  check<2>(256, 384, 512);
  timeit<2>(65536, 256, 2048);
  timeit<2>(16384, 512, 4096);
  timeit<2>(15360, 512, 8192);
  timeit<2>(8192, 8192, 4096);
  g_dma = 2;
  check<2>(256, 384, 512);
  timeit<2>(65536, 256, 2048);
  timeit<2>(16384, 512, 4096);
  timeit<2>(15360, 512, 8192);
  timeit<2>(8192, 8192, 4096);
  g_dma = 0;
  time_split<2>((size_t)16 * 128 * 128 * 128);      // PatchGAN layer-2 input at batch 16 (134 MB)
  time_split<3>((size_t)16 * 128 * 128 * 128);
  time_split<2>((size_t)16 * 32 * 32 * 512);          // layer-4 input (33 MB)
  return 0;
}
