// Prototype (round 4, exploratory -- DESIGN section 6): the data gradient of PatchGAN's wide 4 x 4 convolutions (models/modules/discriminators.py:38-76;
// the layers conv_dgrad32d runs) as an implicit GEMM on the bf16 matrix pipe in 3-product split form, operands PRE-SPLIT into two bf16 planes:
//   dx[n][iy][ix][ci] = sum_(ky, kx, co) dy[n][(iy + p - ky) / S][(ix + p - kx) / S][co] * W[co][ky][kx][ci]        (terms with exact divisions only)
// One launch covers the S * S parity classes (iy % S, ix % S); per class a GEMM of M = N * ceil(H / S) * ceil(W / S) lattice pixels x Cin x
// (k / S)^2 taps * Cout, rows of dy (Cout contiguous) against the transposed filter wt[tap][ci][co] -- the fp32 kernel's scheme
// (cat_amd/csrc/conv_igemm.hip: conv_dgrad32d_kernel<true>) with 2-byte elements: 128 x 128 tile, 2 x 2 waves of 64 x 64, 64 channels per chunk,
// both planes of both operands DMA-ed into XOR-swizzled LDS rows, one barrier per chunk, out-of-image rows fetched beyond num_records (zeros).
// Checks itself against an fp64 evaluation of the definition on a small case, then times the three PatchGAN layers at batch 16.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dgrad_bf16x3.hip -o /tmp/dgrad_bf16x3 && /tmp/dgrad_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;

constexpr int PLANE = 128 * 64 * 2;      // bytes of one 128-row x 64-channel operand plane of a chunk

struct Geom {
  int N, H, W, Cin;        // dx
  int Ho, Wo, Cout;        // dy
  int k, S, p;
  int La, Lb;              // class lattice: ceil(H / S) x ceil(W / S)
  int T;                   // taps per class and axis: k / S
};

__global__ __launch_bounds__(256) void dgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wt, float* __restrict__ dx, const Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  constexpr int BUF = 4 * PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int cls = blockIdx.y, py = cls / g.S, px = cls % g.S;
  const int ntn = g.Cin / 128;
  const int m0 = (blockIdx.x / ntn) * 128, n0 = (blockIdx.x % ntn) * 128;
  const int Mc = g.N * g.La * g.Lb;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wt), 0, 0x7fffffff, 0x00020000);
  const unsigned plA = (unsigned)g.N * g.Ho * g.Wo * g.Cout * 2u, plB = (unsigned)g.k * g.k * g.Cin * g.Cout * 2u;     // bytes between planes
  // staging map: wave w, instruction i -> rows (w * 4 + i) * 8 + (lane >> 3); LDS slot lane & 7 <- source chunk slot ^ (row & 7)
  int rn[4], ra[4], rb[4];
  unsigned cq[4], voA[4], voB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    cq[i] = (unsigned)((lane & 7) ^ (row & 7)) * 16u;
    const int m = m0 + row;
    const bool v = m < Mc;
    const int mm = v ? m : 0;
    rn[i] = v ? mm / (g.La * g.Lb) : -1;
    const int rem = mm - (mm / (g.La * g.Lb)) * (g.La * g.Lb);
    ra[i] = rem / g.Lb;
    rb[i] = rem - ra[i] * g.Lb;
    voB[i] = (unsigned)(n0 + row) * (unsigned)g.Cout * 2u + cq[i];
  }
  // tap t = (ty, tx) of this class: ky = (py + p) % S + S * ty, source row oy = a + (py + p - ky) / S (exact); same along x
  auto locate = [&](int ty, int tx) {
    const int ky = (py + g.p) % g.S + g.S * ty, kx = (px + g.p) % g.S + g.S * tx;
    const int dyo = (py + g.p - ky) / g.S, dxo = (px + g.p - kx) / g.S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = ra[i] + dyo, ox = rb[i] + dxo;
      const bool v = rn[i] >= 0 && (unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo && g.S * ra[i] + py < g.H && g.S * rb[i] + px < g.W;
      voA[i] = v ? (unsigned)((rn[i] * g.Ho + oy) * g.Wo + ox) * (unsigned)g.Cout * 2u + cq[i] : 0x80000000u;
    }
    return ky * g.k + kx;
  };
  int ty = 0, tx = 0, co = 0;               // walk state of the NEXT chunk to fetch (wave-uniform)
  int tap = locate(0, 0);
  auto issue = [&](int buf) {
    unsigned char* d = smem + buf * BUF + wave * 4 * 1024;
    const unsigned soA = (unsigned)co * 2u, soB = (unsigned)(tap * g.Cin) * (unsigned)g.Cout * 2u + (unsigned)co * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_t)(d + p * PLANE + i * 1024), 16, voA[i], (unsigned)p * plA + soA, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_t)(d + (2 + p) * PLANE + i * 1024), 16, voB[i], (unsigned)p * plB + soB, 0, 0);
    }
    co += 64;
    if (co >= g.Cout) {      // next tap (once per Cout / 64 chunks)
      co = 0;
      if (++tx == g.T) {
        tx = 0;
        ++ty;
      }
      if (ty < g.T) tap = locate(ty, tx);
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
      bf8 fa[2][2], fb[2][2];
      const int c = ks * 2 + kg;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int xa = wm * 64 + i * 32 + fr, xb = wn * 64 + i * 32 + fr;
          fa[p][i] = *reinterpret_cast<const bf8*>(s + p * PLANE + xa * 128 + ((c ^ (xa & 7)) << 4));
          fb[p][i] = *reinterpret_cast<const bf8*>(s + (2 + p) * PLANE + xb * 128 + ((c ^ (xb & 7)) << 4));
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
  const int nk = g.T * g.T * (g.Cout / 64);
  issue(0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kc = 0; kc + 1 < nk; ++kc) {
    const int buf = kc & 1;
    issue(buf ^ 1);
    mma(buf);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  mma((nk - 1) & 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (m >= Mc) continue;
      const int n = m / (g.La * g.Lb), rem = m - n * (g.La * g.Lb);
      const int a = rem / g.Lb, b = rem - a * g.Lb;
      const int iy = g.S * a + py, ix = g.S * b + px;
      if (iy >= g.H || ix >= g.W) continue;
      float* o = dx + ((size_t)(n * g.H + iy) * g.W + ix) * g.Cin + n0 + wn * 64 + fr;
#pragma unroll
      for (int j = 0; j < 2; ++j) o[j * 32] = acc[i][j][r];
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
static void split_planes(const std::vector<float>& x, std::vector<bf16_t>& out) {
  out.resize(2 * x.size());
  for (size_t i = 0; i < x.size(); ++i) {
    const bf16_t h = bf16_rne(x[i]);
    out[i] = h;
    out[x.size() + i] = bf16_rne(x[i] - bf16_f32(h));
  }
}

__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed) {      // random-like bf16 bit patterns: timing runs only
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = (bf16_t)((h & 0x80ffu) | (((h >> 16) & 1u ? 0x3f00u : 0x3f80u)));
  }
}

static Geom geom(int N, int H, int W, int Cin, int Cout, int k, int S, int p) {
  Geom g;
  g.N = N, g.H = H, g.W = W, g.Cin = Cin, g.Cout = Cout, g.k = k, g.S = S, g.p = p;
  g.Ho = (H + 2 * p - k) / S + 1, g.Wo = (W + 2 * p - k) / S + 1;
  g.La = (H + S - 1) / S, g.Lb = (W + S - 1) / S, g.T = k / S;
  return g;
}

static void launch(const bf16_t* dy, const bf16_t* wt, float* dx, const Geom& g) {
  const int lds = 2 * 4 * PLANE;
  const int Mc = g.N * g.La * g.Lb;
  const dim3 grid(((Mc + 127) / 128) * (g.Cin / 128), g.S * g.S);
  (void)hipFuncSetAttribute((const void*)dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dgrad_kernel<<<grid, 256, lds>>>(dy, wt, dx, g);
}

static void check(const Geom& g) {
  const size_t ndy = (size_t)g.N * g.Ho * g.Wo * g.Cout, nw = (size_t)g.Cout * g.k * g.k * g.Cin, ndx = (size_t)g.N * g.H * g.W * g.Cin;
  std::vector<float> hdy(ndy), hw(nw), hwt(nw);
  for (auto& v : hdy) v = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 0.8f;
  for (auto& v : hw) v = (float)(rand() & 0xffffff) / 16777216.f * 3.f - 1.2f;
  for (int co = 0; co < g.Cout; ++co)            // W[co][ky][kx][ci] -> wt[tap][ci][co]
    for (int t = 0; t < g.k * g.k; ++t)
      for (int ci = 0; ci < g.Cin; ++ci) hwt[((size_t)t * g.Cin + ci) * g.Cout + co] = hw[((size_t)co * g.k * g.k + t) * g.Cin + ci];
  std::vector<bf16_t> pdy, pwt;
  split_planes(hdy, pdy);
  split_planes(hwt, pwt);
  bf16_t *dy, *wt;
  float* dx;
  (void)hipMalloc(&dy, pdy.size() * 2);
  (void)hipMalloc(&wt, pwt.size() * 2);
  (void)hipMalloc(&dx, ndx * 4);
  (void)hipMemcpy(dy, pdy.data(), pdy.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(wt, pwt.data(), pwt.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemset(dx, 0xff, ndx * 4);            // NaN: an element the kernel does not write shows up
  launch(dy, wt, dx, g);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  std::vector<float> hdx(ndx);
  (void)hipMemcpy(hdx.data(), dx, ndx * 4, hipMemcpyDeviceToHost);
  // the definition, forward-scatter form, fp64
  std::vector<double> ref(ndx, 0.0), mag(ndx, 0.0);
  for (int n = 0; n < g.N; ++n)
    for (int oy = 0; oy < g.Ho; ++oy)
      for (int ox = 0; ox < g.Wo; ++ox)
        for (int ky = 0; ky < g.k; ++ky)
          for (int kx = 0; kx < g.k; ++kx) {
            const int iy = oy * g.S - g.p + ky, ix = ox * g.S - g.p + kx;
            if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
            const float* d = &hdy[((size_t)(n * g.Ho + oy) * g.Wo + ox) * g.Cout];
            double* r = &ref[((size_t)(n * g.H + iy) * g.W + ix) * g.Cin];
            double* a = &mag[((size_t)(n * g.H + iy) * g.W + ix) * g.Cin];
            for (int co = 0; co < g.Cout; ++co) {
              const float* w = &hw[((size_t)co * g.k * g.k + ky * g.k + kx) * g.Cin];
              for (int ci = 0; ci < g.Cin; ++ci) {
                r[ci] += (double)d[co] * w[ci];
                a[ci] += fabs((double)d[co] * w[ci]);
              }
            }
          }
  double worst = 0;
  size_t nan = 0;
  for (size_t i = 0; i < ndx; ++i) {
    if (hdx[i] != hdx[i]) {
      ++nan;
      continue;
    }
    worst = fmax(worst, fabs(hdx[i] - ref[i]) / fmax(mag[i], 1e-30));
  }
  printf("check N %d  %dx%dx%d <- %dx%dx%d  k%d s%d p%d: worst |dx - exact| / sum|terms| = %.3e, unwritten %zu of %zu\n", g.N, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout, g.k,
         g.S, g.p, worst, nan, ndx);
  (void)hipFree(dy);
  (void)hipFree(wt);
  (void)hipFree(dx);
}

static void timeit(const Geom& g, const char* name, int reps = 5) {
  const size_t ndy = (size_t)g.N * g.Ho * g.Wo * g.Cout, nw = (size_t)g.Cout * g.k * g.k * g.Cin, ndx = (size_t)g.N * g.H * g.W * g.Cin;
  bf16_t *dy, *wt;
  float* dx;
  (void)hipMalloc(&dy, ndy * 4);
  (void)hipMalloc(&wt, nw * 4);
  (void)hipMalloc(&dx, ndx * 4);
  fill_kernel<<<2048, 256>>>(dy, 2 * ndy, 3u);
  fill_kernel<<<2048, 256>>>(wt, 2 * nw, 91u);
  launch(dy, wt, dx, g);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch(dy, wt, dx, g);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * g.N * g.Ho * g.Wo * g.Cout * g.k * g.k * g.Cin;      // the convolution's FLOPs, as the library's profile counts them
  if (reps > 5) printf("      (%d back-to-back launches, %.2f s: the clock-settled rate)\n", reps, ms * reps * 1e-3);
  printf("time  %-28s %8.1f us  %6.1f fp32-equivalent TFLOP/s   (conv_dgrad32d in fp32: 0.85 of 157.3 = 134)\n", name, ms * 1e3, flop / ms * 1e-9);
  (void)hipFree(dy);
  (void)hipFree(wt);
  (void)hipFree(dx);
}

int main() {
  srand(5);
  check(geom(2, 16, 16, 128, 128, 4, 2, 1));
  check(geom(1, 12, 20, 128, 192, 4, 1, 1));      // stride 1, ragged lattice (240 rows: a partial tile)
  check(geom(3, 10, 14, 256, 64, 4, 2, 1));       // partial tiles, one chunk per tap
  timeit(geom(16, 128, 128, 128, 256, 4, 2, 1), "D layer 2 (128 <- 256, s2)");
  timeit(geom(16, 64, 64, 256, 512, 4, 2, 1), "D layer 3 (256 <- 512, s2)");
  timeit(geom(16, 32, 32, 512, 512, 4, 1, 1), "D layer 4 (512 <- 512, s1)");
  timeit(geom(16, 64, 64, 256, 512, 4, 2, 1), "D layer 3 (256 <- 512, s2)", 4000);
  return 0;
}
