// Weight gradient of 1 x 1 convolutions over MANY pixels and FEW channels (gfx950): the merged first / second convs of the fused blocks
// (77 -> 60 on 16 x 64 x 64 pixels) and the hidden 1 x 1 layers of the pruned GauGAN student (12 -> 16 ... 36 -> 64 on 4 x 256 x 512).
//
//   dW[co][ci] = sum over pixels p  dy[p][co] * x[p][ci]                 (models/modules/inception_modules.py:135-165 under autograd)
//
// A [C_out x C_in] product of at most a few thousand entries reduced over 10^4 - 10^6 pixels: the implicit-GEMM kernel splits the
// reduction, but its 64 x 64 tiles are mostly padding and it ran these layers at 1 - 4 TFLOP/s = 10 - 15 x their HBM time.  Here the pixel
// stream IS the K dimension and nothing is staged: per 4 pixels a lane quarter q reads pixel p + q, lane r of the quarter one float4 of dy
// (channels 4r .. 4r+3 of a 64-channel group) and one of x -- two fully coalesced 256-byte rows per pixel -- and register j of the dy
// vector with register jj of the x vector are the A / B fragments of v_mfma_f32_16x16x4_f32 for the accumulator tile
// (rows = channels {4r + j}, columns = channels {4c + jj}); a 64 x 64 channel block is 16 tiles, no LDS, no index arithmetic in the loop.
// Eight pixels per wave and step, the next step's vectors requested before the MFMAs of the current one.  A workgroup (4 waves) owns a
// contiguous pixel range, adds its waves' accumulators through LDS and writes ONE partial [C_out][round_up(C_in, 4)]; the shared
// wgrad_reduce kernel sums the partials in a fixed order (deterministic).  blockIdx.y = 64-channel group of x.
#include "common.h"
#include <stdlib.h>

namespace cat_pw {

struct Args {
  const float* x;
  const float* dy;
  float* part;
  int64_t P;          // pixels
  int64_t chunk;      // pixels per workgroup (multiple of 32)
  int xcs, ycs, c4x, c4y, Cout;
};

template <int MG>
__global__ __launch_bounds__(256) void pwgrad_kernel(const Args p) {
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int xc = blockIdx.y * 64 + 4 * r;
  const bool xv = xc < p.c4x;
  const int64_t p0 = (int64_t)blockIdx.x * p.chunk;
  const int64_t p1 = p0 + p.chunk < p.P ? p0 + p.chunk : p.P;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  f4 acc[MG][4][4];
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[g][j][jj] = zero;
  auto load = [&](int64_t pix, f4& xa, f4 (&ya)[MG]) {
    const bool pv = pix < p1;
    xa = (pv && xv) ? *reinterpret_cast<const f4*>(p.x + pix * p.xcs + xc) : zero;
#pragma unroll
    for (int g = 0; g < MG; ++g) {
      const int yc = g * 64 + 4 * r;
      ya[g] = (pv && yc < p.c4y) ? *reinterpret_cast<const f4*>(p.dy + pix * p.ycs + yc) : zero;
    }
  };
  f4 xa[2], ya[2][MG], xn[2], yn[2][MG];
  int64_t base = p0 + 8 * wave + q;
  load(base, xa[0], ya[0]);
  load(base + 4, xa[1], ya[1]);
  for (; base - q - 8 * wave < p1; base += 32) {
    load(base + 32, xn[0], yn[0]);
    load(base + 36, xn[1], yn[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[g][j][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[h][g][j], xa[h][jj], acc[g][j][jj], 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      xa[h] = xn[h];
#pragma unroll
      for (int g = 0; g < MG; ++g) ya[h][g] = yn[h][g];
    }
  }
  // the four waves' accumulators -> one partial (tile by tile through LDS; wave t & 3 owns tile t's sum)
  float* part = p.part + (int64_t)blockIdx.x * p.Cout * p.c4x;
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int t = (g * 4 + j) * 4 + jj;
        *reinterpret_cast<f4*>(red + (wave * 64 + lane) * 4) = acc[g][j][jj];
        __syncthreads();
        if (wave == (t & 3)) {
          f4 s = *reinterpret_cast<const f4*>(red + lane * 4);
#pragma unroll
          for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f4*>(red + (w * 64 + lane) * 4);
          const int ci = blockIdx.y * 64 + 4 * r + jj;      // accumulator layout: register v <-> row 4 * (lane / 16) + v, column lane & 15
          if (ci < p.c4x) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int co = g * 64 + 4 * (4 * q + v) + j;
              if (co < p.Cout) part[(int64_t)co * p.c4x + ci] = s[v];
            }
          }
        }
        __syncthreads();
      }
}

}  // namespace cat_pw

namespace cat {

static int pw_blocks(const cat_conv_t* g) {
  const int64_t P = (int64_t)g->N * g->H * g->W;
  const int groups = (((g->Cin + 3) & ~3) + 63) / 64;
  constexpr int target = 1024;
  int64_t nb = target / groups;               // ~4 workgroups per CU over all channel groups
  const int64_t most = (P + 255) / 256;       // at least 256 pixels (8 steps) per workgroup
  if (nb > most) nb = most;
  return (int)(nb < 1 ? 1 : nb);
}

bool pwgrad_applicable(const cat_conv_t* g) {
  static const int on = getenv("CAT_PWGRAD") ? atoi(getenv("CAT_PWGRAD")) : 1;
  if (!on || g->kh != 1 || g->kw != 1 || g->stride != 1 || g->pad != 0 || g->Ho != g->H || g->Wo != g->W) return false;
  if ((g->xcs & 3) || (g->ycs & 3) || g->Cout > 128 || g->Cin > 256) return false;
  if (g->Cout > 96 && (g->Cin & 127) == 0) return false;          // the direct-to-LDS wide kernel's layers
  return (int64_t)g->N * g->H * g->W >= 8192;                      // few pixels: the general kernel's tiles are not the problem
}

int pwgrad_nblk(const cat_conv_t* g) { return pw_blocks(g); }

// partials [nblk][Cout][round_up(Cin, 4)] into ws; the caller runs the shared reduce
int pwgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s) {
  cat_pw::Args a{};
  a.x = x; a.dy = dy; a.part = ws;
  a.P = (int64_t)g->N * g->H * g->W;
  a.xcs = g->xcs; a.ycs = g->ycs; a.Cout = g->Cout;
  a.c4x = (g->Cin + 3) & ~3;
  a.c4y = (g->Cout + 3) & ~3;
  const int nb = pw_blocks(g);
  a.chunk = ((a.P + nb - 1) / nb + 31) / 32 * 32;
  const dim3 grid(nb, (a.c4x + 63) / 64);
  if (g->Cout <= 64) cat_pw::pwgrad_kernel<1><<<grid, 256, 0, s>>>(a);
  else cat_pw::pwgrad_kernel<2><<<grid, 256, 0, s>>>(a);
  return check_launch("conv2d_pwgrad");
}

}  // namespace cat
