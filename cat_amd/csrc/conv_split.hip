// Opt-in split-bf16 ("bf16x3") forms of the two wide BACKWARD tiles of PatchGAN's 4 x 4 convolutions (models/modules/discriminators.py:38-76: the layers
// conv_dgrad32d / conv_wgrad32d run in exact fp32).  Every fp32 operand is pre-split into two bf16 planes, x = x1 + x2 with x1 = bf16(x),
// x2 = bf16(x - x1) (round to nearest even), and a product is taken as x1 y1 + x1 y2 + x2 y1 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: what is
// dropped is <= 2^-16 of a product.  The host study (tools/bf16x3_numerics.py) shows that form to be a numerical no-op for the two
// backward tiles at the parity suite's bars on the C2, C3 and C4 steps -- and NOT for the forward tile, which therefore has no split form here.
// Kernels as prototyped and checked against fp64 in tools/micro/{dgrad,wgrad}_bf16x3.hip: 128 x 128 workgroup tile, 2 x 2 waves of 64 x 64, 64-deep
// chunks, both planes of both operands DMA-ed straight into LDS (buffer_load_dwordx4 ... lds), one barrier per chunk.
#include "common.h"
#include "../../include/cat_hip_split.h"

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;

namespace cat_split {

constexpr int PLANE = 128 * 64 * 2;      // bytes of one operand plane of a chunk (128 rows x 64 k, or 64 pixels x 128 channels)
constexpr int LDS_BYTES = 2 * 4 * PLANE;

// fp32 -> two bf16 planes [2][n]
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t n4) {
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f4 r = reinterpret_cast<const f4*>(x)[i];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      us4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned u = __float_as_uint(r[e]);
        u += 0x7fffu + ((u >> 16) & 1u);
        h[e] = (unsigned short)(u >> 16);
        r[e] -= __uint_as_float((unsigned)h[e] << 16);
      }
      reinterpret_cast<us4*>(out + (int64_t)p * n4 * 4)[i] = h;
    }
  }
}

struct DGeom {
  int N, H, W, Cin, Ho, Wo, Cout, k, S, p;
  int La, Lb, T;           // class lattice ceil(H / S) x ceil(W / S); taps per class and axis k / S
  int dxcs;
};

// dx[n][iy][ix][ci] = sum_(ky, kx, co) dy[n][(iy + p - ky) / S][(ix + p - kx) / S][co] * w[co][ky][kx][ci]; blockIdx.y = parity class (iy % S, ix % S);
// A rows = dy pixels (Cout contiguous), B rows = wt[ci][tap][co] (cat_conv2d_weight_transpose's layout)
__global__ __launch_bounds__(256) void dgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wt, float* __restrict__ dx, const DGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  constexpr int BUF = 4 * PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int cls = blockIdx.y, py = cls / g.S, px = cls % g.S;
  const int ntn = g.Cin / 128, taps = g.k * g.k;
  const int m0 = (blockIdx.x / ntn) * 128, n0 = (blockIdx.x % ntn) * 128;
  const int Mc = g.N * g.La * g.Lb;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wt), 0, 0x7fffffff, 0x00020000);
  const unsigned plA = (unsigned)g.N * g.Ho * g.Wo * g.Cout * 2u, plB = (unsigned)taps * g.Cin * g.Cout * 2u;     // bytes between planes
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
    voB[i] = (unsigned)(n0 + row) * (unsigned)(taps * g.Cout) * 2u + cq[i];
  }
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
    const unsigned soA = (unsigned)co * 2u, soB = (unsigned)(tap * g.Cout + co) * 2u;
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
      float* o = dx + ((int64_t)(n * g.H + iy) * g.W + ix) * g.dxcs + n0 + wn * 64 + fr;
#pragma unroll
      for (int j = 0; j < 2; ++j) o[j * 32] = acc[i][j][r];
    }
}

struct WGeom {
  int N, H, W, Cin, Ho, Wo, Cout, k, S, p;
  int Wc, R, G;            // chunk: R rows x Wc columns = 64 pixels; G = N * ceil(Ho / R) row groups in all
  int nsplit, gper;
};

// partial dW[split][co][tap][ci] = sum over the split's output rows of dy[n][oy][ox][co] * x[n][oy * S - p + ky][ox * S - p + kx][ci]; K = the pixel index,
// the slow dimension of both operands: [64 pixels][128 channels] tiles as they lie in memory, fragments through ds_read_b64_tr_b16 (result element j of
// lane q of a 16-lane group = element q & 3 at the address lane (q >> 2) + 4 j supplied: tools/micro/ds_read_tr_probe.hip); the 16-byte pieces of pixel
// row r are staged XOR-ed with (r & 3) << 2 so that the four rows of a transposed read leave the same banks
__global__ __launch_bounds__(256) void wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ part, const WGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  typedef __attribute__((address_space(3))) s4v* lds4_t;
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
  int pr[4], pc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kk = (wave * 4 + i) * 4 + (lane >> 4);
    pr[i] = kk / g.Wc;
    pc[i] = kk - pr[i] * g.Wc;
  }
  const unsigned slot = (unsigned)((lane & 15) ^ ((lane >> 4) << 2)) * 16u;
  const int rgroups = (g.Ho + g.R - 1) / g.R;
  int grp = split * g.gper;
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
  const int s = lane & 15, kg = lane >> 5;
  const int toff = (kg * 8 + (s >> 2)) * 256 + (16 * ((lane >> 4) & 1) + 4 * (s & 3)) * 2;
  const int xsw = ((s >> 2) & 3) << 6;
  auto frag = [&](const unsigned char* plane, int ks, int ch) {
    const unsigned char* a = plane + ks * 16 * 256 + ((ch * 2) ^ xsw) + toff;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 4 * 256));
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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
  float* o = part + ((int64_t)split * g.Cout * g.k * g.k) * g.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      float* q = o + ((int64_t)co * g.k * g.k + tap) * g.Cin + ci0 + wn * 64 + fr;
#pragma unroll
      for (int j = 0; j < 2; ++j) q[j * 32] = acc[i][j][r];
    }
}

// dw[co][tap][wcs] (+)= sum over the splits in increasing order (deterministic)
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int64_t rows, int Cin, int wcs,
                                                     int accumulate) {
  const int q4 = Cin >> 2;
  const int64_t total = rows * q4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / q4;
    const int q = (int)(e - row * q4);
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nsplit; ++z) s += *reinterpret_cast<const f4*>(part + ((int64_t)z * rows + row) * Cin + q * 4);
    f4* d = reinterpret_cast<f4*>(dw + row * wcs + q * 4);
    *d = accumulate ? *d + s : s;
  }
}

static bool dgrad_ok(const cat_conv_t* g) {
  const int wcs = g->wcs > 0 ? g->wcs : g->Cin;
  // (the kernels are generic in the kernel size; 4 x 4 at stride 1 | 2 is what tests/test_split_gpu.py has run on hardware, so that is what is admitted)
  return g->pad_mode == CAT_PAD_ZERO && g->kh == 4 && g->kw == 4 && (g->stride == 1 || g->stride == 2) && g->Cin % 128 == 0 && g->Cout % 64 == 0 &&
         g->ycs == g->Cout && wcs == g->Cin && (int64_t)g->N * g->Ho * g->Wo * g->Cout * 4 < (int64_t)2147483647 &&
         (int64_t)g->kh * g->kw * g->Cin * g->Cout * 4 < (int64_t)2147483647;
}

static bool wgrad_ok(const cat_conv_t* g) {
  return g->pad_mode == CAT_PAD_ZERO && g->kh == 4 && g->kw == 4 && (g->stride == 1 || g->stride == 2) && g->Cin % 128 == 0 && g->Cout % 128 == 0 && g->ycs == g->Cout &&
         g->xcs == g->Cin && g->Wo <= 64 &&
         (int64_t)g->N * g->Ho * g->Wo * g->Cout * 4 < (int64_t)2147483647 && (int64_t)g->N * g->H * g->W * g->Cin * 4 < (int64_t)2147483647;
}

static WGeom wgeom(const cat_conv_t* g) {
  WGeom w;
  w.N = g->N, w.H = g->H, w.W = g->W, w.Cin = g->Cin, w.Ho = g->Ho, w.Wo = g->Wo, w.Cout = g->Cout, w.k = g->kh, w.S = g->stride, w.p = g->pad;
  w.Wc = 1;
  while (w.Wc < w.Wo) w.Wc *= 2;
  w.R = 64 / w.Wc;
  w.G = w.N * cat::cdiv(w.Ho, w.R);
  const int tiles = (w.Cin / 128) * (w.Cout / 128) * w.k * w.k;
  int ns = cat::cdiv(512, tiles);
  if (ns > w.G) ns = w.G;
  if (ns < 1) ns = 1;
  w.gper = cat::cdiv(w.G, ns);
  w.nsplit = cat::cdiv(w.G, w.gper);
  return w;
}

}  // namespace cat_split

extern "C" {

int cat_split_bf16(const float* x, void* planes, int64_t n, cat_stream_t stream) {
  CAT_REQUIRE(x && planes && n > 0 && (n & 3) == 0, "split_bf16: n must be a positive multiple of 4");
  const int64_t n4 = n >> 2;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 4096) grid = 4096;
  cat::ProfScope prof("split_bf16", 0.0, 8.0 * (double)n, stream);
  cat_split::split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)planes, n4);
  return cat::check_launch("split_bf16");
}

int cat_conv2d_dgrad_split_applicable(const cat_conv_t* g) { return cat_split::dgrad_ok(g) ? 1 : 0; }

int cat_conv2d_dgrad_split(const cat_conv_t* g, const void* dy_planes, const void* wt_planes, float* dx, int dxcs, cat_stream_t stream) {
  CAT_REQUIRE(cat_split::dgrad_ok(g), "dgrad_split: layer not eligible");
  CAT_REQUIRE(dxcs >= g->Cin, "dgrad_split: output pixel stride");
  cat_split::DGeom d;
  d.N = g->N, d.H = g->H, d.W = g->W, d.Cin = g->Cin, d.Ho = g->Ho, d.Wo = g->Wo, d.Cout = g->Cout, d.k = g->kh, d.S = g->stride, d.p = g->pad;
  d.La = cat::cdiv(g->H, g->stride), d.Lb = cat::cdiv(g->W, g->stride), d.T = g->kh / g->stride, d.dxcs = dxcs;
  static cat::LdsOptIn optin;
  cat::lds_optin(optin, (const void*)cat_split::dgrad_kernel, cat_split::LDS_BYTES);
  const int Mc = d.N * d.La * d.Lb;
  const dim3 grid(cat::cdiv(Mc, 128) * (d.Cin / 128), d.S * d.S);
  cat::ProfScope prof("conv_dgrad_split", 2.0 * g->N * g->Ho * g->Wo * (double)g->Cout * g->kh * g->kw * g->Cin, 0.0, stream);
  cat_split::dgrad_kernel<<<grid, 256, cat_split::LDS_BYTES, (hipStream_t)stream>>>((const bf16_t*)dy_planes, (const bf16_t*)wt_planes, dx, d);
  return cat::check_launch("conv_dgrad_split");
}

int cat_conv2d_wgrad_split_applicable(const cat_conv_t* g) { return cat_split::wgrad_ok(g) ? 1 : 0; }

size_t cat_conv2d_wgrad_split_ws_bytes(const cat_conv_t* g) {
  if (!cat_split::wgrad_ok(g)) return 0;
  const cat_split::WGeom w = cat_split::wgeom(g);
  return (size_t)w.nsplit * g->Cout * g->kh * g->kw * g->Cin * sizeof(float);
}

int cat_conv2d_wgrad_split(const cat_conv_t* g, const void* x_planes, const void* dy_planes, float* dw, int accumulate, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(cat_split::wgrad_ok(g) && ws, "wgrad_split: layer not eligible / no workspace");
  const int wcs = g->wcs > 0 ? g->wcs : g->Cin;
  CAT_REQUIRE(wcs >= g->Cin && (wcs & 3) == 0, "wgrad_split: weight-gradient row stride");
  const cat_split::WGeom w = cat_split::wgeom(g);
  static cat::LdsOptIn optin;
  cat::lds_optin(optin, (const void*)cat_split::wgrad_kernel, cat_split::LDS_BYTES);
  const dim3 grid((w.Cin / 128) * (w.Cout / 128) * w.k * w.k, w.nsplit);
  hipStream_t s = (hipStream_t)stream;
  {
    cat::ProfScope prof("conv_wgrad_split", 2.0 * g->N * g->Ho * g->Wo * (double)g->Cout * g->kh * g->kw * g->Cin, 0.0, stream);
    cat_split::wgrad_kernel<<<grid, 256, cat_split::LDS_BYTES, s>>>((const bf16_t*)x_planes, (const bf16_t*)dy_planes, (float*)ws, w);
  }
  const int64_t rows = (int64_t)g->Cout * g->kh * g->kw;
  int rgrid = (int)((rows * (g->Cin >> 2) + 255) / 256);
  if (rgrid > 4096) rgrid = 4096;
  cat_split::reduce_kernel<<<rgrid, 256, 0, s>>>((const float*)ws, dw, w.nsplit, rows, g->Cin, wcs, accumulate);
  return cat::check_launch("conv_wgrad_split");
}

}  // extern "C"
