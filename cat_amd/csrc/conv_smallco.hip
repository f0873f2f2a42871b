// Convolutions with Cout <= 4 (the generators' 7x7 -> 3 output layer, PatchGAN's 4x4 -> 1 head), stride 1, gfx950.
//
// With 1..4 output channels an MFMA tile would be >= 75 % padding, and fp32 MFMA has no rate advantage over fp32 VALU on
// CDNA4 (both 256 FLOP/clk/CU), so these layers run as direct convolutions on the vector ALU with the input tile (plus halo)
// staged ONCE per channel chunk in LDS -- the k*k im2col re-reads hit LDS instead of L1/L2 -- and the filter taps read through
// the scalar unit (wave-uniform s_load), so the inner loop is 1 ds_read_b128 + 4*Cout v_fma per (tap, channel quad).
//   forward : thread = (output pixel, channel group); LDS tile layout [quad][pixel] -> conflict-free b128 reads
//   wgrad   : thread = (tap, channel quad) accumulates dw over the pixels of many tiles; LDS tile layout [pixel][quad]
// dgrad of these layers has GEMM-N = Cin (large) and stays on the implicit-GEMM kernel.
#include "common.h"
#include <stdlib.h>

namespace cat_smallco {

struct Args {
  const float* x; const float* w; const float* bias; const float* dy; float* y; float* part;
  int N, H, W, Cin, xcs, Ho, Wo, Cout, ycs, kh, kw, pad, reflect, act;
  float slope;
  int cw, c4, tiles_x, tiles_y, nblk, wcs;
  int ksplit, qchunk;   // forward split over input-channel quads: blockIdx.z = slice of qchunk quads, raw sums to part[z][pixel][ycs]
};

__device__ __forceinline__ int src_index(int i, int n, int reflect) {  // -1 = zero padding
  if (reflect) return cat::reflect_idx(i, n);
  return (unsigned)i < (unsigned)n ? i : -1;
}

// Forward.  Workgroup tile = TH rows x 32 columns of output pixels; thread (ty, sx, g) owns the 4 pixels (ty, sx + 8*i) and the
// channel quads q == g (mod CG) of every staged chunk, so one set of filter taps (scalar registers) feeds 4*Cout*4 FMAs and the
// lanes of a wave read consecutive LDS slots.  Staged plane: [CKQ quads][TH+KW-1 rows][40 cols] of float4; the 40-slot row pitch
// (== 8 mod 16) makes the ds_read_b128 of an 8-lane row pair with its neighbours conflict-free.
template <int CO, int TH, int CG, int CKQ, int KW, bool WVEC>
__global__ __launch_bounds__(256) void fwd_kernel(Args p) {
  constexpr int TR = TH + KW - 1, TC = 40, NC = 32 + KW - 1, PLANE = TR * TC;
  static_assert(TH * 8 * CG == 256 && NC <= TC, "tile geometry");
  __shared__ __attribute__((aligned(16))) f4 tile[CKQ * PLANE];
  const int tid = threadIdx.x;
  const int pl = tid % (TH * 8);
  const int g = __builtin_amdgcn_readfirstlane(tid / (TH * 8));  // wave-uniform (TH*8 is a multiple of 64)
  const int ty = pl >> 3, sx = pl & 7;
  const int n = blockIdx.y;
  const int oy0 = (blockIdx.x / p.tiles_x) * TH, ox0 = (blockIdx.x % p.tiles_x) * 32;
  float acc[4][CO];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[i][c] = 0.f;
  const float* xn = p.x + (int64_t)n * p.H * p.W * p.xcs;
  const int qbeg = p.ksplit > 1 ? (int)blockIdx.z * p.qchunk : 0;
  const int nquads = p.ksplit > 1 ? min(p.c4 / 4, qbeg + p.qchunk) : p.c4 / 4;
  for (int q0 = qbeg; q0 < nquads; q0 += CKQ) {
    const int nq = min(CKQ, nquads - q0);
    // stage the chunk: all global loads of a thread are issued before the first LDS store (one latency, not ITERS)
    constexpr int ITERS = (TR * NC * CKQ + 255) / 256;
    f4 buf[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int q = idx % CKQ, pix = idx / CKQ;
      const int r = pix / NC, c = pix - r * NC;
      const int iy = src_index(oy0 - p.pad + r, p.H, p.reflect), ix = src_index(ox0 - p.pad + c, p.W, p.reflect);
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TR * NC * CKQ && q < nq && iy >= 0 && ix >= 0)
        v = *reinterpret_cast<const f4*>(xn + ((int64_t)iy * p.W + ix) * p.xcs + (q0 + q) * 4);
      buf[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int q = idx % CKQ, pix = idx / CKQ;
      const int r = pix / NC, c = pix - r * NC;
      if (idx < TR * NC * CKQ) tile[q * PLANE + r * TC + c] = buf[it];
    }
    __syncthreads();
    for (int q = g; q < nq; q += CG) {
      const int ci = (q0 + q) * 4;
      for (int ky = 0; ky < KW; ++ky) {
        const f4* row = tile + q * PLANE + (ty + ky) * TC + sx;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          f4 xv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = row[kx + 8 * i];
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            const float* wp = p.w + ((int64_t)c * KW * KW + ky * KW + kx) * p.wcs + ci;  // wave-uniform -> scalar loads
            float w0, w1, w2, w3;
            if (WVEC) {
              const f4 wv = *reinterpret_cast<const f4*>(wp);
              w0 = wv[0]; w1 = wv[1]; w2 = wv[2]; w3 = wv[3];
            } else {   // ragged Cin: clamp the index (always in bounds), zero the weight
              w0 = wp[0];
              w1 = wp[ci + 1 < p.Cin ? 1 : 0] * (ci + 1 < p.Cin ? 1.f : 0.f);
              w2 = wp[ci + 2 < p.Cin ? 2 : 0] * (ci + 2 < p.Cin ? 1.f : 0.f);
              w3 = wp[ci + 3 < p.Cin ? 3 : 0] * (ci + 3 < p.Cin ? 1.f : 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[i][c] = fmaf(xv[i][3], w3, fmaf(xv[i][2], w2, fmaf(xv[i][1], w1, fmaf(xv[i][0], w0, acc[i][c]))));
          }
        }
      }
    }
  }
  if (CG > 1) {  // sum the channel groups through LDS (the staged tile is dead by now)
    float* red = reinterpret_cast<float*>(tile);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < CO; ++c) red[(i * CO + c) * 256 + tid] = acc[i][c];
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        float s = 0.f;
        for (int j = 0; j < CG; ++j) s += red[(i * CO + c) * 256 + j * TH * 8 + pl];
        acc[i][c] = s;
      }
  }
  const int oy = oy0 + ty;
  if (oy >= p.Ho) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ox = ox0 + sx + 8 * i;
    if (ox >= p.Wo) continue;
    const int64_t pix = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
    if (p.ksplit > 1) {   // raw partial sums of this channel slice; bias / activation / pad lanes in the shared split-K reduce
      float* po = p.part + ((int64_t)blockIdx.z * p.N * p.Ho * p.Wo + pix) * p.ycs;
#pragma unroll
      for (int c = 0; c < CO; ++c) po[c] = acc[i][c];
      continue;
    }
    float* yo = p.y + pix * p.ycs;
#pragma unroll
    for (int c = 0; c < CO; ++c) yo[c] = cat::apply_act(acc[i][c] + (p.bias ? p.bias[c] : 0.f), p.act, p.slope);
    for (int c = CO; c < p.cw; ++c) yo[c] = 0.f;
  }
}

// wgrad: workgroup (blockIdx.x = tile-stride lane, blockIdx.y = channel chunk of CKQ quads); thread = (tap, quad).
// partial[blockIdx.x][co][tap*c4 + ci] -> reduced by the shared wgrad reduce kernel.
template <int CO, int TILE, int CKQ, int KMAX>
__global__ __launch_bounds__(256) void wgrad_kernel(Args p) {
  constexpr int TI = TILE + KMAX - 1;
  __shared__ __attribute__((aligned(16))) f4 xt[TI * TI * CKQ];
  __shared__ __attribute__((aligned(16))) f4 dyt[TILE * TILE];
  const int tid = threadIdx.x;
  const int taps = p.kh * p.kw;
  const int q = tid % CKQ, tap = tid / CKQ;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int q0 = blockIdx.y * CKQ;
  const int nquads = p.c4 / 4;
  const bool active = tap < taps && q0 + q < nquads;
  const int ti = TILE + p.kh - 1;
  const int ntiles = p.tiles_x * p.tiles_y * p.N;
  f4 acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / (p.tiles_x * p.tiles_y), tt = t - n * (p.tiles_x * p.tiles_y);
    const int oy0 = (tt / p.tiles_x) * TILE, ox0 = (tt % p.tiles_x) * TILE;
    const float* xn = p.x + (int64_t)n * p.H * p.W * p.xcs;
    constexpr int ITERS = (TI * TI * CKQ + 255) / 256;
    f4 buf[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int qq = idx % CKQ, pix = idx / CKQ;
      const int r = pix / ti, c = pix - r * ti;
      const int iy = src_index(oy0 - p.pad + r, p.H, p.reflect), ix = src_index(ox0 - p.pad + c, p.W, p.reflect);
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < ti * ti * CKQ && q0 + qq < nquads && iy >= 0 && ix >= 0)
        v = *reinterpret_cast<const f4*>(xn + ((int64_t)iy * p.W + ix) * p.xcs + (q0 + qq) * 4);
      buf[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int qq = idx % CKQ, pix = idx / CKQ;
      const int r = pix / ti, c = pix - r * ti;
      if (idx < ti * ti * CKQ) xt[(r * TI + c) * CKQ + qq] = buf[it];
    }
    for (int idx = tid; idx < TILE * TILE; idx += 256) {
      const int oy = oy0 + idx / TILE, ox = ox0 + idx % TILE;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < p.Ho && ox < p.Wo) v = *reinterpret_cast<const f4*>(p.dy + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ycs);
      dyt[idx] = v;  // channels >= Cout of dy are zero padding
    }
    __syncthreads();
    if (active) {
      for (int py = 0; py < TILE; ++py) {
#pragma unroll 4
        for (int px = 0; px < TILE; ++px) {
          const f4 xv = xt[((py + ky) * TI + px + kx) * CKQ + q];
          const f4 g = dyt[py * TILE + px];
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[c] += xv * g[c];
        }
      }
    }
  }
  if (active) {
    const int K = taps * p.c4;
#pragma unroll
    for (int c = 0; c < CO; ++c)
      if (c < p.Cout)
        *reinterpret_cast<f4*>(p.part + ((int64_t)blockIdx.x * p.Cout + c) * K + tap * p.c4 + (q0 + q) * 4) = acc[c];
  }
}


// dgrad of a 4x4 / stride 2 / pad 1 conv with 3 or 6 INPUT channels (PatchGAN's first layer: the gradient into the image).
// GEMM-N = Cin would be >= 62 % MFMA padding, and the implicit-GEMM kernel walks the taps in the outer K loop, so its four shifted reads
// of dy miss L2 (measured 18x the compulsory fetch).  Here dy is staged ONCE per 16-channel chunk in LDS and every output pixel class
// (parity of iy + pad, ix + pad) is one wave: class (py, px) at coarse position (a, b) is
//     dx[2a + py - pad][2b + px - pad][ci] = sum_{jy, jx in {0,1}} sum_co dy[a - jy][b - jx][co] * w[co][py + 2 jy][px + 2 jx][ci]
// Thread (ty, sx) of a class owns the 8 coarse pixels (ty + 8 u, sx + 8 i), u < 2, i < 4.  The class's filter taps of the chunk sit in LDS
// as [tap][quad][co-in-quad][CI] and are read as broadcast b128 (scalar loads would share lgkmcnt with the LDS reads and serialise).
template <int CI, int CKQ>
__global__ __launch_bounds__(256) void dgrad_s2_kernel(Args p) {
  constexpr int TH = 16, TR = TH + 2, NC = 34, TC = 40, PLANE = TR * TC;
  constexpr int WQ = 4 * CI;                     // floats per (tap, quad): 4 output channels of the forward conv x CI
  static_assert(CKQ == 4 && (WQ & 3) == 0, "filter stage: one (class, tap, quad, co) row per thread");
  __shared__ __attribute__((aligned(16))) f4 tile[CKQ * PLANE];
  __shared__ __attribute__((aligned(16))) float wl[4 * 4 * CKQ * WQ];
  const int tid = threadIdx.x;
  const int pl = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = cls >> 1, px = cls & 1;
  const int ty = pl >> 3, sx = pl & 7;
  const int n = blockIdx.y;
  const int a0 = (blockIdx.x / p.tiles_x) * TH, b0 = (blockIdx.x % p.tiles_x) * 32;
  // first coarse row / column of this class whose output pixel is >= 0, and that pixel's coordinate
  const int amy = p.pad > py ? (p.pad - py + 1) >> 1 : 0, amx = p.pad > px ? (p.pad - px + 1) >> 1 : 0;
  const int fy = 2 * amy + py - p.pad, fx = 2 * amx + px - p.pad;
  float acc[8][CI];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[i][c] = 0.f;
  const float* dyn = p.dy + (int64_t)n * p.Ho * p.Wo * p.ycs;
  const int nquads = p.c4 / 4;       // c4 = round_up(Cout, 4)
  // filter stage map: thread -> (class, tap, quad, co-in-quad)
  const int we = tid & 3, wq = (tid >> 2) & 3, wt = (tid >> 4) & 3, wc = tid >> 6;
  const int wky = (wc >> 1) + 2 * (wt >> 1), wkx = (wc & 1) + 2 * (wt & 1);
  for (int q0 = 0; q0 < nquads; q0 += CKQ) {
    const int nq = min(CKQ, nquads - q0);
    constexpr int ITERS = (TR * NC * CKQ + 255) / 256;
    f4 buf[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int q = idx % CKQ, pix = idx / CKQ;
      const int r = pix / NC, c = pix - r * NC;
      const int oy = a0 - 1 + r, ox = b0 - 1 + c;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TR * NC * CKQ && q < nq && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo)
        v = *reinterpret_cast<const f4*>(dyn + ((int64_t)oy * p.Wo + ox) * p.ycs + (q0 + q) * 4);
      buf[it] = v;
    }
    float wr[CI];
    {
      const int co = (q0 + wq) * 4 + we;
      const float* wp = p.w + ((int64_t)min(co, p.Cout - 1) * 16 + wky * 4 + wkx) * p.wcs;
#pragma unroll
      for (int c = 0; c < CI; ++c) wr[c] = co < p.Cout ? wp[c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int q = idx % CKQ, pix = idx / CKQ;
      const int r = pix / NC, c = pix - r * NC;
      if (idx < TR * NC * CKQ) tile[q * PLANE + r * TC + c] = buf[it];
    }
#pragma unroll
    for (int c = 0; c < CI; ++c) wl[((wc * 4 + wt) * CKQ + wq) * WQ + we * CI + c] = wr[c];
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int jy = t >> 1, jx = t & 1;
        const f4* wrow = reinterpret_cast<const f4*>(wl + ((cls * 4 + t) * CKQ + q) * WQ);
        float wv[WQ];
#pragma unroll
        for (int k = 0; k < WQ / 4; ++k) {
          const f4 v = wrow[k];
          wv[4 * k] = v[0]; wv[4 * k + 1] = v[1]; wv[4 * k + 2] = v[2]; wv[4 * k + 3] = v[3];
        }
        const f4* row = tile + q * PLANE + (ty + amy - jy + 1) * TC + sx + amx - jx + 1;
        f4 xv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[u * 4 + i] = row[u * 8 * TC + 8 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][c] = fmaf(xv[i][e], wv[e * CI + c], acc[i][c]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int iy = 2 * (a0 + ty + 8 * u) + fy;
    if (iy >= p.H) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ix = 2 * (b0 + sx + 8 * i) + fx;
      if (ix >= p.W) continue;
      float* o = p.y + (((int64_t)n * p.H + iy) * p.W + ix) * p.xcs;
#pragma unroll
      for (int c = 0; c < CI; ++c) o[c] = acc[u * 4 + i][c];
      for (int c = CI; c < p.cw; ++c) o[c] = 0.f;
    }
  }
}

}  // namespace cat_smallco

namespace cat {

bool smallco_applicable(const cat_conv_t* g) {  // the layers that exist in CAT's networks: 7x7 -> 3 (generators), 4x4 -> 1 (PatchGAN)
  return (g->Cout == 1 || g->Cout == 3) && g->stride == 1 && g->kh == g->kw && (g->kh == 4 || g->kh == 7);
}

static cat_smallco::Args make_args(const cat_conv_t* g) {
  cat_smallco::Args a{};
  a.N = g->N; a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.xcs = g->xcs; a.Ho = g->Ho; a.Wo = g->Wo; a.Cout = g->Cout; a.ycs = g->ycs;
  a.kh = g->kh; a.kw = g->kw; a.pad = g->pad; a.reflect = g->pad_mode == CAT_PAD_REFLECT; a.act = g->act; a.slope = g->slope;
  a.c4 = (g->Cin + 3) & ~3;
  a.wcs = g->wcs > 0 ? g->wcs : g->Cin;
  return a;
}

// small images with many channels (PatchGAN head) -> 8x8 tiles, 4 channel groups, 64-channel chunks; else 16x16 tiles
static bool small_image_cfg(const cat_conv_t* g) {
  return g->Cin >= 64 && (int64_t)g->N * cdiv(g->Ho, 16) * cdiv(g->Wo, 16) < 512 && g->kh <= 4;
}

template <int CO, int KW, bool WVEC>
static void launch_fwd(cat_smallco::Args& a, const cat_conv_t* g, bool small_image, hipStream_t s) {
  a.tiles_x = cdiv(g->Wo, 32);
  if (small_image) {   // few pixels, many channels: 8-row tiles, 4 channel groups
    a.tiles_y = cdiv(g->Ho, 8);
    cat_smallco::fwd_kernel<CO, 8, 4, 8, KW, WVEC><<<dim3(a.tiles_x * a.tiles_y, g->N, a.ksplit > 1 ? a.ksplit : 1), 256, 0, s>>>(a);
  } else {
    a.tiles_y = cdiv(g->Ho, 32);
    cat_smallco::fwd_kernel<CO, 32, 1, 2, KW, WVEC><<<dim3(a.tiles_x * a.tiles_y, g->N), 256, 0, s>>>(a);
  }
}

template <int CO, int KW>
static void launch_fwd_v(cat_smallco::Args& a, const cat_conv_t* g, bool sm, hipStream_t s) {
  if ((a.wcs & 3) == 0) launch_fwd<CO, KW, true>(a, g, sm, s);
  else launch_fwd<CO, KW, false>(a, g, sm, s);
}

static bool fwd_small_image(const cat_conv_t* g) { return g->Cin >= 64 && (int64_t)g->N * cdiv(g->Ho, 32) * cdiv(g->Wo, 32) < 256; }

// PatchGAN's head at batch 16 is 64 workgroups walking 32 channel chunks each: with a workspace the channel range is cut into
// slices of >= 2 chunks (8 quads each) until ~512 workgroups exist
int smallco_fwd_ksplit(const cat_conv_t* g) {
  constexpr int on = 1;
  if (!on || !fwd_small_image(g)) return 1;
  const int blocks = g->N * cdiv(g->Ho, 8) * cdiv(g->Wo, 32);
  const int chunks = cdiv(((g->Cin + 3) & ~3) / 4, 8);
  int ks = cdiv(512, blocks);
  if (ks > chunks / 2) ks = chunks / 2;
  if (ks < 2) return 1;
  const int per = cdiv(chunks, ks);
  ks = cdiv(chunks, per);
  return ks < 2 ? 1 : ks;
}

int smallco_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, float* ws, int ksplit, hipStream_t s) {
  cat_smallco::Args a = make_args(g);
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.cw = g->ycw > g->Cout ? g->ycw : g->Cout;
  const bool sm = fwd_small_image(g);
  a.ksplit = 1;
  if (sm && ws && ksplit > 1) {
    const int chunks = cdiv(((g->Cin + 3) & ~3) / 4, 8);
    a.ksplit = ksplit;
    a.qchunk = cdiv(chunks, ksplit) * 8;
    a.part = ws;
  }
  if (g->Cout == 1 && g->kh == 4) launch_fwd_v<1, 4>(a, g, sm, s);
  else if (g->Cout == 1 && g->kh == 7) launch_fwd_v<1, 7>(a, g, sm, s);
  else if (g->Cout == 3 && g->kh == 4) launch_fwd_v<3, 4>(a, g, sm, s);
  else if (g->Cout == 3 && g->kh == 7) launch_fwd_v<3, 7>(a, g, sm, s);
  else { set_error("smallco fwd: (Cout=%d, k=%d) has no specialisation", g->Cout, g->kh); return -22; }
  return check_launch("conv2d_fwd_smallco");
}

int smallco_wgrad_nblk(const cat_conv_t* g) {
  const bool sm = small_image_cfg(g);
  const int tile = sm ? 8 : 16;
  const int ntiles = g->N * cdiv(g->Ho, tile) * cdiv(g->Wo, tile);
  const int ckq = sm ? 16 : 4;
  const int chunks = cdiv(((g->Cin + 3) & ~3) / 4, ckq);
  int nb = cdiv(1024, chunks);
  if (nb > ntiles) nb = ntiles;
  if (nb > 256) nb = 256;
  return nb < 1 ? 1 : nb;
}

// writes partials [nblk][Cout][taps*c4] into ws; the caller runs the shared reduce
int smallco_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s) {
  cat_smallco::Args a = make_args(g);
  a.x = x; a.dy = dy; a.part = ws;
  const bool sm = small_image_cfg(g);
  const int nb = smallco_wgrad_nblk(g);
  a.nblk = nb;
  if (sm) {
    a.tiles_x = cdiv(g->Wo, 8); a.tiles_y = cdiv(g->Ho, 8);
    cat_smallco::wgrad_kernel<4, 8, 16, 4><<<dim3(nb, cdiv(a.c4 / 4, 16)), 256, 0, s>>>(a);
  } else {
    a.tiles_x = cdiv(g->Wo, 16); a.tiles_y = cdiv(g->Ho, 16);
    cat_smallco::wgrad_kernel<4, 16, 4, 7><<<dim3(nb, cdiv(a.c4 / 4, 4)), 256, 0, s>>>(a);
  }
  return check_launch("conv2d_wgrad_smallco");
}


// dgrad of PatchGAN's first layer (see dgrad_s2_kernel): 4x4, stride 2, zero pad 1, 3 or 6 input channels
bool smallci_dgrad_applicable(const cat_conv_t* g) {
  static const int on = getenv("CAT_SMALLCI_DGRAD") ? atoi(getenv("CAT_SMALLCI_DGRAD")) : 1;
  const int wcs = g->wcs > 0 ? g->wcs : g->Cin;
  return on && (g->Cin == 3 || g->Cin == 6) && g->stride == 2 && g->kh == 4 && g->kw == 4 && g->pad == 1 && g->pad_mode == CAT_PAD_ZERO &&
         g->Cout >= 16 && wcs >= g->Cin && (g->ycs & 3) == 0;
}

// dx: [N, H, W, dxcs], channels [Cin, dxcw) zero-filled
int smallci_dgrad(const cat_conv_t* g, const float* dy, const float* w, float* dx, int dxcs, int dxcw, hipStream_t s) {
  cat_smallco::Args a = make_args(g);
  a.dy = dy; a.w = w; a.y = dx;
  a.xcs = dxcs;
  a.cw = dxcw > g->Cin ? dxcw : g->Cin;
  a.c4 = (g->Cout + 3) & ~3;
  // coarse class grid: every class has at most ceil(H / 2) x ceil(W / 2) pixels
  a.tiles_x = cdiv(cdiv(g->W, 2), 32);
  a.tiles_y = cdiv(cdiv(g->H, 2), 16);
  const dim3 grid(a.tiles_x * a.tiles_y, g->N);
  if (g->Cin == 6) cat_smallco::dgrad_s2_kernel<6, 4><<<grid, 256, 0, s>>>(a);
  else cat_smallco::dgrad_s2_kernel<3, 4><<<grid, 256, 0, s>>>(a);
  return check_launch("conv2d_dgrad_smallci");
}

}  // namespace cat
