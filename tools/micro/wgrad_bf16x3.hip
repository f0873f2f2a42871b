// Prototype (round 4, exploratory -- DESIGN section 6): the weight gradient of PatchGAN's wide 4 x 4 convolutions (the layers conv_wgrad32d runs) on the
// bf16 matrix pipe in 3-product split form, operands PRE-SPLIT into two bf16 planes:
//   dW[co][ky][kx][ci] = sum_(n, oy, ox) dy[n][oy][ox][co] * x[n][oy * S - p + ky][ox * S - p + kx][ci]
// A GEMM whose K dimension -- the pixel index -- is the SLOW dimension of both operands: the tiles are DMA-ed into LDS as they lie in memory
// ([64 pixels][128 channels], 256-byte rows) and the MFMA fragments (8 consecutive k per lane) come out of ds_read_b64_tr_b16, whose lane map
// tools/micro/ds_read_tr_probe.hip measured: result element j of lane q (of a 16-lane group) = element q & 3 at the address lane (q >> 2) + 4 j supplied.
// One workgroup = 128 output channels x 128 input channels of ONE tap over a slice of the output rows (split K; partial sums [split][Cout][k * k][Cin],
// reduced by a second pass as the fp32 kernel's are); a chunk = R output rows x Wc columns = 64 pixels (Wc = the power of two >= Wo).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wgrad_bf16x3.hip -o /tmp/wgrad_bf16x3 && /tmp/wgrad_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;

constexpr int PLANE = 64 * 128 * 2;      // bytes of one 64-pixel x 128-channel operand plane of a chunk

struct Geom {
  int N, H, W, Cin, Ho, Wo, Cout, k, S, p;
  int Wc, R, G;            // chunk: R rows x Wc columns; G = N * ceil(Ho / R) row groups in all
  int nsplit, gper;        // row groups per split
};

template <bool SWZ>
__global__ __launch_bounds__(256) void wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ part, const Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  typedef __attribute__((address_space(3))) s4* lds4_t;
  constexpr int BUF = 4 * PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nci = g.Cin / 128, nco = g.Cout / 128;
  int b = blockIdx.x;
  const int ci0 = (b % nci) * 128;
  b /= nci;
  const int co0 = (b % nco) * 128;
  const int tap = b / nco, ky = tap / g.k, kx = tap - ky * g.k;
  const int split = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, 0x7fffffff, 0x00020000);
  const unsigned plX = (unsigned)g.N * g.H * g.W * g.Cin * 2u, plD = (unsigned)g.N * g.Ho * g.Wo * g.Cout * 2u;
  // staging map: wave w, instruction i -> pixel rows (w * 4 + i) * 4 + (lane >> 4) of the chunk, 16-byte slot lane & 15 (8 channels)
  int pr[4], pc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kk = (wave * 4 + i) * 4 + (lane >> 4);
    pr[i] = kk / g.Wc;
    pc[i] = kk - pr[i] * g.Wc;
  }
  // LDS slot lane & 15 of pixel row r holds the row's 16-byte piece (lane & 15) ^ ((r & 3) << 2): the four pixel rows one transposed read gathers
  // (256 bytes = one bank row apart) then sit in four different 32-byte blocks, and the two 16-lane groups of a 32-lane pass in the odd / even ones
  const unsigned slot = (unsigned)((lane & 15) ^ (SWZ ? (lane >> 4) << 2 : 0)) * 16u;
  const int rgroups = (g.Ho + g.R - 1) / g.R;
  int grp = split * g.gper;                      // walk state of the NEXT chunk to fetch
  const int gend = min(g.G, grp + g.gper);
  int n = grp / rgroups, oy0 = (grp - n * rgroups) * g.R;
  auto issue = [&](int buf) {
    unsigned char* d = smem + buf * BUF + wave * 4 * 1024;
    unsigned voD[4], voX[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = oy0 + pr[i], ox = pc[i];
      const bool v = oy < g.Ho && ox < g.Wo;
      const int iy = oy * g.S - g.p + ky, ix = ox * g.S - g.p + kx;
      const bool vx = v && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
      voD[i] = v ? ((unsigned)((n * g.Ho + oy) * g.Wo + ox) * (unsigned)g.Cout + (unsigned)co0) * 2u + slot : 0x80000000u;
      voX[i] = vx ? ((unsigned)((n * g.H + iy) * g.W + ix) * (unsigned)g.Cin + (unsigned)ci0) * 2u + slot : 0x80000000u;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rD, (lds_t)(d + p * PLANE + i * 1024), 16, voD[i], (unsigned)p * plD, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_t)(d + (2 + p) * PLANE + i * 1024), 16, voX[i], (unsigned)p * plX, 0, 0);
    }
    ++grp;
    oy0 += g.R;
    if (oy0 >= g.Ho) {
      oy0 = 0;
      ++n;
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment read: lane l, s = l & 15 supplies the address of pixel row (s >> 2) (+ 4 for the second read), channels 16 * ((l >> 4) & 1) + 4 * (s & 3) ..+3
  const int s = lane & 15, kg = lane >> 5;
  const int toff = (kg * 8 + (s >> 2)) * 256 + (16 * ((lane >> 4) & 1) + 4 * (s & 3)) * 2;
  const int xsw = SWZ ? ((s >> 2) & 3) << 6 : 0;
  auto frag = [&](const unsigned char* plane, int ks, int ch) {
    const unsigned char* a = plane + ks * 16 * 256 + ((ch * 2) ^ xsw) + toff;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 4 * 256));
    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8, v);
  };
  auto mma = [&](int buf) {
    const unsigned char* sb = smem + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf8 fa[2][2], fb[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[p][i] = frag(sb + p * PLANE, ks, wm * 64 + i * 32);
          fb[p][i] = frag(sb + (2 + p) * PLANE, ks, wn * 64 + i * 32);
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
  const int nk = gend - split * g.gper;
  if (nk > 0) {
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
  }
  const int fr = lane & 31;
  float* o = part + ((size_t)split * g.Cout * g.k * g.k) * g.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      float* q = o + ((size_t)co * g.k * g.k + tap) * g.Cin + ci0 + wn * 64 + fr;
#pragma unroll
      for (int j = 0; j < 2; ++j) q[j * 32] = acc[i][j][r];
    }
}

static bf16_t bf16_rne(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static float bf16_f32(bf16_t h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static void split_planes(const std::vector<float>& v, std::vector<bf16_t>& out) {
  out.resize(2 * v.size());
  for (size_t i = 0; i < v.size(); ++i) {
    const bf16_t h = bf16_rne(v[i]);
    out[i] = h;
    out[v.size() + i] = bf16_rne(v[i] - bf16_f32(h));
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

static Geom geom(int N, int H, int W, int Cin, int Cout, int k, int S, int p, int nsplit) {
  Geom g;
  g.N = N, g.H = H, g.W = W, g.Cin = Cin, g.Cout = Cout, g.k = k, g.S = S, g.p = p;
  g.Ho = (H + 2 * p - k) / S + 1, g.Wo = (W + 2 * p - k) / S + 1;
  g.Wc = 1;
  while (g.Wc < g.Wo) g.Wc *= 2;
  if (g.Wc > 64) {
    printf("Wo > 64 needs column chunks: not in this prototype\n");
    exit(1);
  }
  g.R = 64 / g.Wc;
  g.G = N * ((g.Ho + g.R - 1) / g.R);
  g.nsplit = nsplit < g.G ? nsplit : g.G;
  g.gper = (g.G + g.nsplit - 1) / g.nsplit;
  g.nsplit = (g.G + g.gper - 1) / g.gper;
  return g;
}

static bool g_swz = false;      // XOR-swizzled LDS rows (see the staging map)
static void launch(const bf16_t* x, const bf16_t* dy, float* part, const Geom& g) {
  const int lds = 2 * 4 * PLANE;
  const dim3 grid((g.Cin / 128) * (g.Cout / 128) * g.k * g.k, g.nsplit);
  if (g_swz) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    wgrad_kernel<true><<<grid, 256, lds>>>(x, dy, part, g);
  } else {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    wgrad_kernel<false><<<grid, 256, lds>>>(x, dy, part, g);
  }
}

static void check(const Geom& g) {
  const size_t nx = (size_t)g.N * g.H * g.W * g.Cin, ndy = (size_t)g.N * g.Ho * g.Wo * g.Cout, nw = (size_t)g.Cout * g.k * g.k * g.Cin;
  std::vector<float> hx(nx), hdy(ndy);
  for (auto& v : hx) v = (float)(rand() & 0xffffff) / 16777216.f * 2.f - 0.8f;
  for (auto& v : hdy) v = (float)(rand() & 0xffffff) / 16777216.f * 3.f - 1.2f;
  std::vector<bf16_t> px, pdy;
  split_planes(hx, px);
  split_planes(hdy, pdy);
  bf16_t *x, *dy;
  float* part;
  (void)hipMalloc(&x, px.size() * 2);
  (void)hipMalloc(&dy, pdy.size() * 2);
  (void)hipMalloc(&part, (size_t)g.nsplit * nw * 4);
  (void)hipMemcpy(x, px.data(), px.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dy, pdy.data(), pdy.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemset(part, 0xff, (size_t)g.nsplit * nw * 4);
  launch(x, dy, part, g);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  std::vector<float> hp((size_t)g.nsplit * nw);
  (void)hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost);
  std::vector<double> ref(nw, 0.0), mag(nw, 0.0);
  for (int n = 0; n < g.N; ++n)
    for (int oy = 0; oy < g.Ho; ++oy)
      for (int ox = 0; ox < g.Wo; ++ox)
        for (int ky = 0; ky < g.k; ++ky)
          for (int kx = 0; kx < g.k; ++kx) {
            const int iy = oy * g.S - g.p + ky, ix = ox * g.S - g.p + kx;
            if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
            const float* d = &hdy[((size_t)(n * g.Ho + oy) * g.Wo + ox) * g.Cout];
            const float* xv = &hx[((size_t)(n * g.H + iy) * g.W + ix) * g.Cin];
            for (int co = 0; co < g.Cout; ++co) {
              double* r = &ref[((size_t)co * g.k * g.k + ky * g.k + kx) * g.Cin];
              double* a = &mag[((size_t)co * g.k * g.k + ky * g.k + kx) * g.Cin];
              for (int ci = 0; ci < g.Cin; ++ci) {
                r[ci] += (double)d[co] * xv[ci];
                a[ci] += fabs((double)d[co] * xv[ci]);
              }
            }
          }
  double worst = 0;
  size_t nan = 0;
  for (size_t i = 0; i < nw; ++i) {
    double sum = 0;
    for (int sp = 0; sp < g.nsplit; ++sp) {
      const float v = hp[(size_t)sp * nw + i];
      if (v != v) ++nan;
      sum += v;
    }
    worst = fmax(worst, fabs(sum - ref[i]) / fmax(mag[i], 1e-30));
  }
  printf("check%s N %d  x %dx%dx%d  dy %dx%dx%d  k%d s%d p%d, chunk %d x %d, %d splits: worst |dW - exact| / sum|terms| = %.3e, unwritten %zu\n", g_swz ? " (swizzled)" : "", g.N, g.H, g.W, g.Cin, g.Ho,
         g.Wo, g.Cout, g.k, g.S, g.p, g.R, g.Wc, g.nsplit, worst, nan);
  (void)hipFree(x);
  (void)hipFree(dy);
  (void)hipFree(part);
}

static void timeit(const Geom& g, const char* name) {
  const size_t nx = (size_t)g.N * g.H * g.W * g.Cin, ndy = (size_t)g.N * g.Ho * g.Wo * g.Cout, nw = (size_t)g.Cout * g.k * g.k * g.Cin;
  bf16_t *x, *dy;
  float* part;
  (void)hipMalloc(&x, nx * 4);
  (void)hipMalloc(&dy, ndy * 4);
  (void)hipMalloc(&part, (size_t)g.nsplit * nw * 4);
  fill_kernel<<<2048, 256>>>(x, 2 * nx, 3u);
  fill_kernel<<<2048, 256>>>(dy, 2 * ndy, 91u);
  launch(x, dy, part, g);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 5;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch(x, dy, part, g);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * g.N * g.Ho * g.Wo * g.Cout * g.k * g.k * g.Cin;
  printf("time%s  %-28s %d splits, %5d workgroups: %8.1f us  %6.1f fp32-equivalent TFLOP/s   (conv_wgrad32d in fp32: 0.80 of 157.3 = 126; partial sums reduced separately in both)\n",
         g_swz ? " (swizzled)" : "", name, g.nsplit, (g.Cin / 128) * (g.Cout / 128) * g.k * g.k * g.nsplit, ms * 1e3, flop / ms * 1e-9);
  (void)hipFree(x);
  (void)hipFree(dy);
  (void)hipFree(part);
}

int main() {
  srand(5);
  check(geom(2, 16, 16, 128, 128, 4, 2, 1, 2));      // 8 x 8 output: one chunk per image
  check(geom(1, 12, 20, 128, 256, 4, 1, 1, 3));      // stride 1, 11 x 19 output: chunks of 2 x 32 with ragged rows and columns
  check(geom(3, 20, 70, 256, 128, 4, 2, 1, 4));      // 10 x 35 output: chunks of 1 x 64
  timeit(geom(16, 128, 128, 128, 256, 4, 2, 1, 16), "D layer 2 (256 x 128, s2)");
  timeit(geom(16, 64, 64, 256, 512, 4, 2, 1, 4), "D layer 3 (512 x 256, s2)");
  timeit(geom(16, 32, 32, 512, 512, 4, 1, 1, 2), "D layer 4 (512 x 512, s1)");
  g_swz = true;
  check(geom(2, 16, 16, 128, 128, 4, 2, 1, 2));
  check(geom(1, 12, 20, 128, 256, 4, 1, 1, 3));
  check(geom(3, 20, 70, 256, 128, 4, 2, 1, 4));
  timeit(geom(16, 128, 128, 128, 256, 4, 2, 1, 16), "D layer 2 (256 x 128, s2)");
  timeit(geom(16, 64, 64, 256, 512, 4, 2, 1, 4), "D layer 3 (512 x 256, s2)");
  timeit(geom(16, 32, 32, 512, 512, 4, 1, 1, 2), "D layer 4 (512 x 512, s1)");
  return 0;
}
