// LDS-tile forward convolution for narrow outputs (Cout <= 48), stride 1, "same" 3x3 / 5x5, NHWC, gfx950.
//
// OPT-IN (CAT_CONV_TILE=1) until it has been validated and timed on hardware: written at the end of round 1 without GPU time left;
// tools/debug/emulate_conv_tile.py replays its staging / fragment indexing in numpy against a direct convolution.
//
// Why: with GEMM-N = 12..48 the im2col kernel (conv_igemm.hip) re-fetches its A chunk from L2 for each of the k*k taps -- 12.8 FLOP
// per L2 byte at N = 18 (student 77 -> 18, 5x5: 39 TFLOP/s) and 1.07 GB of fabric traffic per launch on the teacher's 256 -> 42 5x5
// layer.  Here the (8 + k - 1) x (32 + k - 1) input patch of a workgroup's 8 x 32 output pixels is staged ONCE per
// 16-channel chunk as [row][col][16 ch + 4 pad] floats; the MFMA A fragment of (16 consecutive pixels, tap, 16 channels) is then one
// ds_read_b128 at a compile-time offset from a per-lane base (pixel pitch 20 floats: the 16 lanes of a quarter wave hit 16 disjoint
// 4-bank groups), feeding 4 v_mfma_f32_16x16x4_f32 per 16 output channels.  B fragments ([co][tap][ci] filter rows, 64 B per output
// channel and tap) come straight from L1/L2, prefetched one tap ahead.  K order = channel chunk outer, taps inner.
//   workgroup: 256 threads = 4 waves; wave w owns output rows 2w, 2w+1 of the tile = 4 M-tiles of 16 pixels; NT = ceil(Cout / 16).
#include "common.h"
#include <stdlib.h>

namespace cat_tile {

struct Args {
  const float* x; const float* w; const float* bias; float* y;
  int N, H, W, Cin, xcs, Cout, ycs, pad, reflect, act;
  float slope;
  int cw, c4, wcs, tiles_x, tiles_y;
};

__device__ __attribute__((aligned(16))) float g_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int TH = 8, TW = 32, PITCH = 20, CK = 16;

template <int KS, int NT>
__global__ __launch_bounds__(256) void fwd_kernel(Args p) {
  constexpr int TR = TH + KS - 1, TC = TW + KS - 1, TAPS = KS * KS;
  constexpr int SLOTS = TR * TC * 4;              // float4 per staged chunk
  constexpr int ITERS = (SLOTS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float tile[TR * TC * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.y;
  const int oy0 = (blockIdx.x / p.tiles_x) * TH, ox0 = (blockIdx.x % p.tiles_x) * TW;

  // staging map: slot = (tile pixel, channel quad); the pixel's offset in x (or "invalid") does not depend on the chunk
  unsigned soff[ITERS];
  bool sval[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * 256;
    const int pix = idx >> 2, quad = idx & 3;
    const int r = pix / TC, c = pix - r * TC;
    int iy = oy0 - p.pad + r, ix = ox0 - p.pad + c;
    bool v = idx < SLOTS;
    if (p.reflect) {
      v = v && iy >= -p.pad && iy < p.H + p.pad && ix >= -p.pad && ix < p.W + p.pad;
      iy = cat::reflect_idx(iy, p.H);
      ix = cat::reflect_idx(ix, p.W);
    } else {
      v = v && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    }
    sval[it] = v;
    soff[it] = v ? (unsigned)(((n * p.H + iy) * p.W + ix) * p.xcs + quad * 4) : 0u;   // < 2^32 elements (checked on the host)
  }
  f4 sreg[ITERS];
  auto gload = [&](int c0) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int quad = (tid + it * 256) & 3;
      const bool v = sval[it] && c0 + quad * 4 < p.c4;
      sreg[it] = *reinterpret_cast<const f4*>(v ? p.x + soff[it] + c0 : g_zero);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      if (idx < SLOTS) *reinterpret_cast<f4*>(tile + (idx >> 2) * PITCH + (idx & 3) * 4) = sreg[it];
    }
  };

  // B rows of this lane: output channel j*16 + lr, channel quad lq of the chunk
  const float* brow[NT];
  bool bval[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = j * 16 + lr;
    bval[j] = co < p.Cout;
    brow[j] = p.w + (int64_t)(bval[j] ? co : 0) * TAPS * p.wcs + lq * 4;
  }

  f4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const float* abase = tile + ((2 * wave) * TC + lr) * PITCH + lq * 4;
  const int nch = (p.c4 + CK - 1) / CK;
  gload(0);
  for (int ch = 0; ch < nch; ++ch) {
    const int c0 = ch * CK;
    __syncthreads();   // every wave is done reading the previous chunk
    sstore();
    __syncthreads();
    if (ch + 1 < nch) gload(c0 + CK);   // in flight behind this chunk's MFMA stream
    const float* pb[NT];
    int incb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bool v = bval[j] && c0 + lq * 4 < p.c4;
      pb[j] = v ? brow[j] + c0 : g_zero;
      incb[j] = v ? p.wcs : 0;
    }
    f4 fb[NT], fbn[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f4*>(pb[j]);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
      if (tap + 1 < TAPS) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          pb[j] += incb[j];
          fbn[j] = *reinterpret_cast<const f4*>(pb[j]);
        }
      }
      f4 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f4*>(abase + (((i >> 1) + ky) * TC + (i & 1) * 16 + kx) * PITCH);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
      if (tap + 1 < TAPS) {
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = fbn[j];
      }
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = oy0 + 2 * wave + (i >> 1);
    if (oy >= p.H) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ox = ox0 + (i & 1) * 16 + lq * 4 + rg;
      if (ox >= p.W) continue;
      float* yo = p.y + (((int64_t)n * p.H + oy) * p.W + ox) * p.ycs;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = j * 16 + lr;
        if (co < p.Cout) yo[co] = cat::apply_act(acc[i][j][rg] + (p.bias ? p.bias[co] : 0.f), p.act, p.slope);
        else if (co < p.cw) yo[co] = 0.f;
      }
    }
  }
}

}  // namespace cat_tile

namespace cat {

bool conv_tile_applicable(const cat_conv_t* g) {
  static const int on = getenv("CAT_CONV_TILE") ? atoi(getenv("CAT_CONV_TILE")) : 0;
  if (!on) return false;
  const int wcs = g->wcs > 0 ? g->wcs : g->Cin;
  const int cw = g->ycw > g->Cout ? g->ycw : g->Cout;
  const int nt = cdiv(g->Cout, 16);
  return g->stride == 1 && g->kh == g->kw && (g->kh == 3 || g->kh == 5) && g->pad == (g->kh - 1) / 2 && g->Cout <= 48 && cw <= nt * 16 &&
         (wcs & 3) == 0 && (on >= 2 || (int64_t)g->N * cdiv(g->H, cat_tile::TH) * cdiv(g->W, cat_tile::TW) >= 128) &&   // 2: force (tests)
         (int64_t)g->N * g->H * g->W * g->xcs < (int64_t)4294967295LL;
}

int conv_tile_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
  cat_tile::Args a{};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.N = g->N; a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.xcs = g->xcs; a.Cout = g->Cout; a.ycs = g->ycs;
  a.pad = g->pad; a.reflect = g->pad_mode == CAT_PAD_REFLECT; a.act = g->act; a.slope = g->slope;
  a.cw = g->ycw > g->Cout ? g->ycw : g->Cout;
  a.c4 = (g->Cin + 3) & ~3;
  a.wcs = g->wcs > 0 ? g->wcs : g->Cin;
  a.tiles_x = cdiv(g->W, cat_tile::TW);
  a.tiles_y = cdiv(g->H, cat_tile::TH);
  const dim3 grid(a.tiles_x * a.tiles_y, g->N);
  const int nt = cdiv(g->Cout, 16);
#define CAT_TILE_LAUNCH(KS, NT) cat_tile::fwd_kernel<KS, NT><<<grid, 256, 0, s>>>(a)
  if (g->kh == 3) {
    if (nt == 1) CAT_TILE_LAUNCH(3, 1); else if (nt == 2) CAT_TILE_LAUNCH(3, 2); else CAT_TILE_LAUNCH(3, 3);
  } else {
    if (nt == 1) CAT_TILE_LAUNCH(5, 1); else if (nt == 2) CAT_TILE_LAUNCH(5, 2); else CAT_TILE_LAUNCH(5, 3);
  }
#undef CAT_TILE_LAUNCH
  return check_launch("conv2d_fwd_tile");
}

}  // namespace cat
