// LDS-tile convolution for narrow outputs (GEMM-N <= 48), stride 1, 3x3 / 5x5, NHWC, gfx950: forward ("same" padding) and dgrad.
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
//   workgroup: 256 threads = 4 waves; wave w owns output rows 2w, 2w+1 of the tile = 4 M-tiles of 16 pixels; NT = ceil(N / 16).
// One kernel body, two operand conventions (out[oy][ox][n] = sum_{ky,kx,c} in[oy - padv + ky][ox - padv + kx][c] * B(tap, c, n)):
//   forward : in = x (c = input channel), n = output channel, B = w[n][tap][c] (a float4 of c per lane), padv = pad, zero / reflect
//   dgrad   : in = dy (c = OUTPUT channel of the forward conv), n = input channel, B = w[c][TAPS-1-tap][n] (four scalars per lane,
//             c-stride TAPS*wcs), padv = k - 1 - pad_eff, zero padding only; the output plane is x's (zero pad) or the padded plane
//             of a reflect-padded conv (pad_eff = 0: the caller folds the border back, as for conv_dgrad_kernel).
#include "common.h"
#include <stdlib.h>

namespace cat_tile {

struct Args {
  const float* x; const float* w; const float* bias; float* y;   // x = staged operand (x or dy), y = output (y or dx)
  int N, H, W, xcs;        // staged tensor: plane size, pixel stride
  int Ho, Wo, ycs;         // output plane, pixel stride
  int Ck, Nn;              // reduction channels (valid), output channels (valid)
  int pad, reflect, act;   // pad = padv of the header comment
  float slope;
  int cw, c4, wcs, tiles_x, tiles_y;   // c4 = round_up(Ck, 4)
};

__device__ __attribute__((aligned(16))) float g_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int TH = 8, TW = 32, PITCH = 20, CK = 16;

template <int KS, int NT, bool DGRAD>
__global__ __launch_bounds__(256) void tile_kernel(Args p) {
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

  // B operand of this lane: output channel n = j*16 + lr, reduction channels c0 + lq*4 .. +3 of the chunk
  //   forward: one float4 at w[n][tap][c0 + lq*4];  dgrad: four scalars w[c0 + lq*4 + t][TAPS-1-tap][n]
  constexpr int BE = DGRAD ? 4 : 1;
  const float* brow[NT];
  bool bval[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nn = j * 16 + lr;
    bval[j] = nn < p.Nn;
    if (DGRAD) brow[j] = p.w + ((int64_t)(lq * 4) * TAPS + (TAPS - 1)) * p.wcs + (bval[j] ? nn : 0);
    else brow[j] = p.w + (int64_t)(bval[j] ? nn : 0) * TAPS * p.wcs + lq * 4;
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
    const float* pb[NT][BE];
    int incb[NT][BE];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int e = 0; e < BE; ++e) {
        if (DGRAD) {   // rows c >= Ck do not exist in w: per-element validity (only the last quad of the last chunk is ragged)
          const bool v = bval[j] && c0 + lq * 4 + e < p.Ck;
          pb[j][e] = v ? brow[j] + (int64_t)(c0 + e) * TAPS * p.wcs : g_zero;
          incb[j][e] = v ? -p.wcs : 0;
        } else {
          const bool v = bval[j] && c0 + lq * 4 < p.c4;
          pb[j][e] = v ? brow[j] + c0 : g_zero;
          incb[j][e] = v ? p.wcs : 0;
        }
      }
    }
    auto bload = [&](int j) {
      f4 v;
      if (DGRAD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = *pb[j][e];
      } else {
        v = *reinterpret_cast<const f4*>(pb[j][0]);
      }
      return v;
    };
    f4 fb[NT], fbn[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = bload(j);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
      if (tap + 1 < TAPS) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int e = 0; e < BE; ++e) pb[j][e] += incb[j][e];
          fbn[j] = bload(j);
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
    if (oy >= p.Ho) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ox = ox0 + (i & 1) * 16 + lq * 4 + rg;
      if (ox >= p.Wo) continue;
      float* yo = p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ycs;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = j * 16 + lr;
        if (co < p.Nn) yo[co] = cat::apply_act(acc[i][j][rg] + (p.bias ? p.bias[co] : 0.f), p.act, p.slope);
        else if (co < p.cw) yo[co] = 0.f;
      }
    }
  }
}

}  // namespace cat_tile

namespace cat {

static int tile_switch() {
  static const int on = getenv("CAT_CONV_TILE") ? atoi(getenv("CAT_CONV_TILE")) : 0;   // 0 off, 1 on, 2 on for any size (tests)
  return on;
}

static bool tile_geometry_ok(const cat_conv_t* g, int nn, int cw, int64_t out_pixels_h, int64_t out_pixels_w, int64_t staged_elems) {
  const int on = tile_switch();
  if (!on) return false;
  const int nt = cdiv(nn, 16);
  return g->stride == 1 && g->kh == g->kw && (g->kh == 3 || g->kh == 5) && g->pad == (g->kh - 1) / 2 && nn <= 48 && cw <= nt * 16 &&
         (on >= 2 || (int64_t)g->N * cdiv(out_pixels_h, cat_tile::TH) * cdiv(out_pixels_w, cat_tile::TW) >= 128) &&
         staged_elems < (int64_t)4294967295LL;
}

bool conv_tile_applicable(const cat_conv_t* g) {
  const int wcs = g->wcs > 0 ? g->wcs : g->Cin;
  const int cw = g->ycw > g->Cout ? g->ycw : g->Cout;
  return (wcs & 3) == 0 && tile_geometry_ok(g, g->Cout, cw, g->H, g->W, (int64_t)g->N * g->H * g->W * g->xcs);
}

// dgrad: GEMM-N = Cin; the output plane is x's (zero padding) or the reflect-padded plane (the caller folds it)
bool conv_tile_dgrad_applicable(const cat_conv_t* g, int dxcw) {
  const bool refl = g->pad_mode == CAT_PAD_REFLECT;
  const int hin = refl ? g->H + 2 * g->pad : g->H, win = refl ? g->W + 2 * g->pad : g->W;
  const int cw = dxcw > g->Cin ? dxcw : g->Cin;
  return g->Cin <= 32 && tile_geometry_ok(g, g->Cin, cw, hin, win, (int64_t)g->N * g->Ho * g->Wo * g->ycs);   // N-tiles: 1 or 2 (registers)
}

template <bool DGRAD>
static void tile_launch(const cat_tile::Args& a, int ks, int nt, dim3 grid, hipStream_t s) {
#define CAT_TILE_LAUNCH(KS, NT) cat_tile::tile_kernel<KS, NT, DGRAD><<<grid, 256, 0, s>>>(a)
  if (ks == 3) {
    if (nt == 1) CAT_TILE_LAUNCH(3, 1); else if (nt == 2) CAT_TILE_LAUNCH(3, 2); else if constexpr (!DGRAD) CAT_TILE_LAUNCH(3, 3);
  } else {
    if (nt == 1) CAT_TILE_LAUNCH(5, 1); else if (nt == 2) CAT_TILE_LAUNCH(5, 2); else if constexpr (!DGRAD) CAT_TILE_LAUNCH(5, 3);
  }
#undef CAT_TILE_LAUNCH
}

int conv_tile_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
  cat_tile::Args a{};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.N = g->N; a.H = g->H; a.W = g->W; a.xcs = g->xcs; a.Ho = g->H; a.Wo = g->W; a.ycs = g->ycs;
  a.Ck = g->Cin; a.Nn = g->Cout;
  a.pad = g->pad; a.reflect = g->pad_mode == CAT_PAD_REFLECT; a.act = g->act; a.slope = g->slope;
  a.cw = g->ycw > g->Cout ? g->ycw : g->Cout;
  a.c4 = (g->Cin + 3) & ~3;
  a.wcs = g->wcs > 0 ? g->wcs : g->Cin;
  a.tiles_x = cdiv(a.Wo, cat_tile::TW);
  a.tiles_y = cdiv(a.Ho, cat_tile::TH);
  tile_launch<false>(a, g->kh, cdiv(g->Cout, 16), dim3(a.tiles_x * a.tiles_y, g->N), s);
  return check_launch("conv2d_fwd_tile");
}

int conv_tile_dgrad(const cat_conv_t* g, const float* dy, const float* w, const float* bias, float* dx, int dxcs, int dxcw, hipStream_t s) {
  const bool refl = g->pad_mode == CAT_PAD_REFLECT;
  cat_tile::Args a{};
  a.x = dy; a.w = w; a.bias = bias; a.y = dx;
  a.N = g->N; a.H = g->Ho; a.W = g->Wo; a.xcs = g->ycs;
  a.Ho = refl ? g->H + 2 * g->pad : g->H; a.Wo = refl ? g->W + 2 * g->pad : g->W; a.ycs = dxcs;
  a.Ck = g->Cout; a.Nn = g->Cin;
  a.pad = g->kh - 1 - (refl ? 0 : g->pad); a.reflect = 0; a.act = g->act; a.slope = g->slope;
  a.cw = dxcw > g->Cin ? dxcw : g->Cin;
  a.c4 = (g->Cout + 3) & ~3;
  a.wcs = g->wcs > 0 ? g->wcs : g->Cin;
  a.tiles_x = cdiv(a.Wo, cat_tile::TW);
  a.tiles_y = cdiv(a.Ho, cat_tile::TH);
  tile_launch<true>(a, g->kh, cdiv(g->Cin, 16), dim3(a.tiles_x * a.tiles_y, g->N), s);
  return check_launch("conv2d_dgrad_tile");
}

}  // namespace cat

// ------------------------------------------------------------------------------------------------ wgrad from LDS tiles
// dw[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci] for the narrow student layers, whose im2col wgrad spends more vector-ALU time
// on the per-pixel gather than the matrix pipe spends on the product (77 -> 18, 5x5: 34 TFLOP/s).  Per workgroup: one pair of 16-wide
// output-channel tiles x one 16-channel chunk of x x a strided subset of the 8 x 32 pixel tiles.  Both operands of a tile are staged in
// LDS at a pixel pitch of exactly 16 floats -- an MFMA fragment (16 channels x 4 consecutive pixels) is then 64 consecutive floats,
// read with one conflict-free ds_read_b32: A = dy[4 px][16 co], B = x[4 px shifted by the tap][16 ci], one v_mfma_f32_16x16x4_f32 per
// (pixel group, tap, co tile).  The k*k taps are dealt round-robin to the four waves (accumulators: <= 7 taps x 2 co tiles), every wave
// walks all 64 pixel groups of the tile.  Raw partial sums go to ws[z][co][tap*c4 + ci]; the shared wgrad_reduce_kernel finishes.
namespace cat_tile {

struct WArgs {
  const float* x; const float* dy; float* part;
  int N, H, W, xcs, ycs, Cin, Cout, pad, reflect;
  int c4, K, tiles_x, tiles_y, ntiles;
};

template <int KS, int NCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_tile_kernel(WArgs p) {
  constexpr int TR = TH + KS - 1, TC = TW + KS - 1, TAPS = KS * KS;
  constexpr int XSLOTS = TR * TC * 4, XITERS = (XSLOTS + 255) / 256;
  constexpr int DSLOTS = TH * TW * 4 * NCO, DITERS = DSLOTS / 256;
  constexpr int NSLOT = (TAPS + 3) / 4;   // taps per wave (round-robin)
  __shared__ __attribute__((aligned(16))) float xt[TR * TC * 16];
  __shared__ __attribute__((aligned(16))) float dyt[NCO * TH * TW * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int co0 = blockIdx.y * 16 * NCO, c0 = blockIdx.z * 16;
  const int cout4 = (p.Cout + 3) & ~3;

  f4 xreg[XITERS], dreg[DITERS];
  auto gload = [&](int t) {
    const int per = p.tiles_x * p.tiles_y;
    const int n = t / per, tt = t - n * per;
    const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
#pragma unroll
    for (int it = 0; it < XITERS; ++it) {
      const int idx = tid + it * 256;
      const int pix = idx >> 2, quad = idx & 3;
      const int r = pix / TC, c = pix - r * TC;
      int iy = oy0 - p.pad + r, ix = ox0 - p.pad + c;
      bool v = idx < XSLOTS && c0 + quad * 4 < p.c4;
      if (p.reflect) {
        v = v && iy >= -p.pad && iy < p.H + p.pad && ix >= -p.pad && ix < p.W + p.pad;
        iy = cat::reflect_idx(iy, p.H);
        ix = cat::reflect_idx(ix, p.W);
      } else {
        v = v && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      }
      xreg[it] = *reinterpret_cast<const f4*>(v ? p.x + (((int64_t)n * p.H + iy) * p.W + ix) * p.xcs + c0 + quad * 4 : g_zero);
    }
#pragma unroll
    for (int it = 0; it < DITERS; ++it) {
      const int idx = tid + it * 256;
      const int pix = idx / (4 * NCO), quad = idx - pix * (4 * NCO);
      const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
      const bool v = oy < p.H && ox < p.W && co0 + quad * 4 < cout4;
      dreg[it] = *reinterpret_cast<const f4*>(v ? p.dy + (((int64_t)n * p.H + oy) * p.W + ox) * p.ycs + co0 + quad * 4 : g_zero);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int it = 0; it < XITERS; ++it) {
      const int idx = tid + it * 256;
      if (idx < XSLOTS) *reinterpret_cast<f4*>(xt + idx * 4) = xreg[it];           // [pixel][16]: slot idx = pixel * 4 + quad
    }
#pragma unroll
    for (int it = 0; it < DITERS; ++it) {
      const int idx = tid + it * 256;
      const int pix = idx / (4 * NCO), quad = idx - pix * (4 * NCO);
      *reinterpret_cast<f4*>(dyt + ((quad >> 2) * TH * TW + pix) * 16 + (quad & 3) * 4) = dreg[it];   // [co tile][pixel][16]
    }
  };

  f4 acc[NSLOT][NCO];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s)
#pragma unroll
    for (int j = 0; j < NCO; ++j) acc[s][j] = f4{0.f, 0.f, 0.f, 0.f};

  static_assert(TAPS == 4 * (NSLOT - 1) + 1, "tap dealing assumes k*k = 1 (mod 4)");
  int toff[NSLOT];   // LDS offset of the tap each accumulator slot of this wave owns
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int tap = min(wave + 4 * s, TAPS - 1);
    toff[s] = ((tap / KS) * TC + tap % KS) * 16;
  }
  int t = blockIdx.x;
  if (t < p.ntiles) gload(t);
  for (; t < p.ntiles; t += gridDim.x) {
    __syncthreads();   // previous tile fully consumed
    sstore();
    __syncthreads();
    if (t + (int)gridDim.x < p.ntiles) gload(t + gridDim.x);
    for (int r = 0; r < TH; ++r) {
#pragma unroll
      for (int cg = 0; cg < TW / 4; ++cg) {
        float a[NCO];
#pragma unroll
        for (int j = 0; j < NCO; ++j) a[j] = dyt[(j * TH * TW + r * TW + cg * 4 + lq) * 16 + lr];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          // taps wave, wave + 4, ...: k*k = 4 * (NSLOT - 1) + 1, so only the last slot is conditional (it exists for wave 0 alone)
          if (s + 1 < NSLOT || wave == 0) {
            const float b = xt[toff[s] + (r * TC + cg * 4 + lq) * 16 + lr];
#pragma unroll
            for (int j = 0; j < NCO; ++j) acc[s][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[s][j], 0, 0, 0);
          }
        }
      }
    }
  }

  float* part = p.part + (int64_t)blockIdx.x * p.Cout * p.K;
  const int ci = c0 + lr;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int tap = wave + 4 * s;
    if (tap >= TAPS || ci >= p.c4) continue;
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = co0 + j * 16 + lq * 4 + rg;
        if (co < p.Cout) part[(int64_t)co * p.K + tap * p.c4 + ci] = acc[s][j][rg];
      }
    }
  }
}

}  // namespace cat_tile

namespace cat {

bool conv_tile_wgrad_applicable(const cat_conv_t* g) {
  const int on = tile_switch();
  if (!on) return false;
  const int lo = g->Cin < g->Cout ? g->Cin : g->Cout;
  return g->stride == 1 && g->kh == g->kw && (g->kh == 3 || g->kh == 5) && g->pad == (g->kh - 1) / 2 && lo <= 32 && g->Cin <= 128 &&
         g->Cout <= 128 && (on >= 2 || (int64_t)g->N * cdiv(g->H, cat_tile::TH) * cdiv(g->W, cat_tile::TW) >= 64);
}

// pixel-split count: enough workgroups for ~2 per CU, never more than there are tiles
int conv_tile_wgrad_nsplit(const cat_conv_t* g) {
  const int nco = g->Cout <= 16 ? 1 : 2;
  const int groups = cdiv(g->Cout, 16 * nco) * cdiv((g->Cin + 3) & ~3, 16);
  const int ntiles = g->N * cdiv(g->H, cat_tile::TH) * cdiv(g->W, cat_tile::TW);
  int nz = cdiv(512, groups);
  if (nz > ntiles) nz = ntiles;
  if (nz > 256) nz = 256;
  return nz < 1 ? 1 : nz;
}

// partial sums -> ws[nsplit][Cout][taps * c4]; the caller runs wgrad_reduce_kernel over them
int conv_tile_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s) {
  cat_tile::WArgs a{};
  a.x = x; a.dy = dy; a.part = ws;
  a.N = g->N; a.H = g->H; a.W = g->W; a.xcs = g->xcs; a.ycs = g->ycs; a.Cin = g->Cin; a.Cout = g->Cout;
  a.pad = g->pad; a.reflect = g->pad_mode == CAT_PAD_REFLECT;
  a.c4 = (g->Cin + 3) & ~3;
  a.K = g->kh * g->kw * a.c4;
  a.tiles_x = cdiv(g->W, cat_tile::TW);
  a.tiles_y = cdiv(g->H, cat_tile::TH);
  a.ntiles = g->N * a.tiles_x * a.tiles_y;
  const int nco = g->Cout <= 16 ? 1 : 2;
  const dim3 grid(conv_tile_wgrad_nsplit(g), cdiv(g->Cout, 16 * nco), cdiv(a.c4, 16));
  if (g->kh == 3) {
    if (nco == 1) cat_tile::wgrad_tile_kernel<3, 1><<<grid, 256, 0, s>>>(a); else cat_tile::wgrad_tile_kernel<3, 2><<<grid, 256, 0, s>>>(a);
  } else {
    if (nco == 1) cat_tile::wgrad_tile_kernel<5, 1><<<grid, 256, 0, s>>>(a); else cat_tile::wgrad_tile_kernel<5, 2><<<grid, 256, 0, s>>>(a);
  }
  return check_launch("conv2d_wgrad_tile");
}

}  // namespace cat
