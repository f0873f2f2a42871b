// Weight gradient of stride-1 "same" 3x3 / 5x5 convolutions whose channel counts are small on one side (the pruned student's hidden
// layers: 77 <-> 7..18 channels on 64 x 64 planes), gfx950.
//
//   dW[co][ky][kx][ci] = sum over pixels  dy[n][oy][ox][co] * x[n][oy + ky - pad][ox + kx - pad][ci]
//
// The implicit-GEMM wgrad kernel gathers every x element once per tap (25 x for a 5x5) with per-pixel index arithmetic on the vector
// ALU and runs these layers at 27-34 TFLOP/s.  Here a workgroup walks 8 x 8 pixel tiles: the x patch (tile + halo) and the dy tile are
// staged ONCE in LDS and every tap reads the patch at an immediate offset.  GEMM view per tile: M = output channels (rows from dy),
// N = input channels (columns from x), K = the tile's 64 pixels (four pixels per v_mfma_f32_16x16x4_f32: lane quarter <-> pixel,
// lane & 15 <-> channel, so both fragments are conflict-free ds_read_b32).  The accumulator tiles (tap, ci tile, co tile) are dealt
// round-robin to the four waves at compile time (125 tiles of a 77 -> 15 5x5: 32 / 31 / 31 / 31), so every LDS address is
// lane base + immediate and the k loop is straight-line MFMA + ds_read.  Each workgroup keeps its accumulators over all its tiles
// and writes ONE partial dW; the shared wgrad_reduce kernel sums the partials (deterministic order).
#include "common.h"
#include <stdlib.h>

namespace cat_tw {

constexpr int TH = 8, TW = 8, NPIX = TH * TW;

constexpr int pitch_of(int tiles) { return tiles == 1 ? 16 : (tiles == 2 ? 48 : 80); }   // floats per staged pixel; pitch % 64 in {16, 48}:
//                                                                                     the four lane quarters of a ds_read_b32 hit disjoint banks

struct Args {
  const float* x;
  const float* dy;
  float* part;
  int N, H, W, Cin, xcs, Cout, ycs, pad, reflect;
  int c4x, c4y;                // channel extents staged (multiples of 4)
  int tiles_x, tiles_img, ntiles;
  int K;                       // taps * c4x: row length of a partial
};

// the MFMA stream of wave WV over one staged tile: two k-steps per tile row (columns 0..3 and 4..7).  Operands of the NEXT k-step are
// requested before the MFMAs of the current one are issued (two register sets in ping-pong, scheduling fences between the phases): with
// one wave per SIMD nothing else hides the LDS latency.  Every address is lane base + immediate.
template <int PT, int QT, int KS, int WV, int NU, int H>
__device__ __forceinline__ void tw_load(float (&b)[NU], float (&a)[QT], const float* xl, const float* yl) {
  constexpr int XP = pitch_of(PT), YP = pitch_of(QT), PW = TW + KS - 1;
  constexpr int U = KS * KS * PT * QT;
#pragma unroll
  for (int q = 0; q < QT; ++q) a[q] = yl[H * 4 * YP + 16 * q];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int u = WV + 4 * i;
    if (u < U) {
      const int pt = (u / QT) % PT, tap = u / (QT * PT);
      const int ky = tap / KS, kx = tap - ky * KS;
      b[i] = xl[(ky * PW + kx + H * 4) * XP + 16 * pt];
    }
  }
}

template <int PT, int QT, int KS, int WV, int NU>
__device__ __forceinline__ void tw_mma(f4 (&acc)[NU], const float (&b)[NU], const float (&a)[QT]) {
  constexpr int U = KS * KS * PT * QT;
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int u = WV + 4 * i;
    if (u < U) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u % QT], b[i], acc[i], 0, 0, 0);
  }
}

template <int PT, int QT, int KS, int WV, int NU>
__device__ __forceinline__ void mma_tile(f4 (&acc)[NU], const float* xl, const float* yl) {
  constexpr int XP = pitch_of(PT), YP = pitch_of(QT), PW = TW + KS - 1;
  float b0[NU], b1[NU], a0[QT], a1[QT];
  tw_load<PT, QT, KS, WV, NU, 0>(b0, a0, xl, yl);
#pragma unroll 1
  for (int r = 0; r < TH; ++r) {
    tw_load<PT, QT, KS, WV, NU, 1>(b1, a1, xl, yl);
    __builtin_amdgcn_sched_barrier(0);
    tw_mma<PT, QT, KS, WV, NU>(acc, b0, a0);
    __builtin_amdgcn_sched_barrier(0);
    xl += PW * XP;
    yl += TW * YP;
    if (r + 1 < TH) tw_load<PT, QT, KS, WV, NU, 0>(b0, a0, xl, yl);
    __builtin_amdgcn_sched_barrier(0);
    tw_mma<PT, QT, KS, WV, NU>(acc, b1, a1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int PT, int QT, int KS, int WV, int NU>
__device__ __forceinline__ void store_partial(const f4 (&acc)[NU], const Args& p, float* part, int lr, int lq) {
  constexpr int U = KS * KS * PT * QT;
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int u = WV + 4 * i;
    if (u < U) {
      const int qt = u % QT, pt = (u / QT) % PT, tap = u / (QT * PT);
      const int ci = 16 * pt + lr;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = 16 * qt + lq * 4 + rg;
        if (co < p.Cout && ci < p.c4x) part[(int64_t)co * p.K + tap * p.c4x + ci] = acc[i][rg];
      }
    }
  }
}

template <int PT, int QT, int KS>
__global__ __launch_bounds__(256) void twgrad_kernel(Args p) {
  constexpr int XP = pitch_of(PT), YP = pitch_of(QT), PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int XQ = XP / 4, YQ = YP / 4;
  constexpr int XSLOTS = PH * PW * XQ, YSLOTS = NPIX * YQ;
  constexpr int XIT = (XSLOTS + 255) / 256, YIT = (YSLOTS + 255) / 256;
  constexpr int XF = XIT * 1024, YF = YIT * 1024;     // floats per staged operand (whole 1 KB DMA rows: the tail row is padding)
  constexpr int U = KS * KS * PT * QT, NU = (U + 3) / 4;
  typedef __attribute__((address_space(3))) void* lds_t;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][x patch | dy tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, 0x7fffffff, 0x00020000);

  f4 acc[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};

  // staging map (the same for every tile): DMA instruction it of wave w moves slots [(it * 4 + w) * 64, + 64) -- one contiguous KB of the
  // LDS image; slot -> (patch pixel, channel quad).  Packed per lane: row | col << 8 | quad << 16, or -1 for a slot that stays zero.
  int xmap[XIT], ymap[YIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int slot = (it * 4 + wave) * 64 + lane;
    const int q = slot % XQ, pix = slot / XQ;
    const int r = pix / PW, c = pix - r * PW;
    xmap[it] = (slot < XSLOTS && q * 4 < p.c4x) ? (r | (c << 8) | (q << 16)) : -1;
  }
#pragma unroll
  for (int it = 0; it < YIT; ++it) {
    const int slot = (it * 4 + wave) * 64 + lane;
    const int q = slot % YQ, pix = slot / YQ;
    ymap[it] = (slot < YSLOTS && q * 4 < p.c4y) ? ((pix / TW) | ((pix % TW) << 8) | (q << 16)) : -1;
  }
  // direct-to-LDS staging of tile t into buffer buf: lanes outside the plane / the channel extent carry an out-of-range offset (the buffer
  // unit then writes zeros: zero padding, and dy = 0 for pixels beyond the plane, which therefore add nothing)
  auto stage = [&](int t, int buf) {
    const int n = t / p.tiles_img, tt = t - n * p.tiles_img;
    const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
    float* xb = smem + buf * (XF + YF);
    float* yb = xb + XF;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int m = xmap[it];
      int iy = oy0 - p.pad + (m & 255), ix = ox0 - p.pad + ((m >> 8) & 255);
      bool v = m >= 0;
      if (p.reflect) {
        v = v && iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
        iy = cat::reflect_idx(iy, p.H);
        ix = cat::reflect_idx(ix, p.W);
      } else {
        v = v && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      }
      const unsigned vo = v ? ((unsigned)((n * p.H + iy) * p.W + ix) * (unsigned)p.xcs + (unsigned)(m >> 16) * 4u) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_t)(xb + (it * 4 + wave) * 256), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int m = ymap[it];
      const int oy = oy0 + (m & 255), ox = ox0 + ((m >> 8) & 255);
      const bool v = m >= 0 && oy < p.H && ox < p.W;
      const unsigned vo = v ? ((unsigned)((n * p.H + oy) * p.W + ox) * (unsigned)p.ycs + (unsigned)(m >> 16) * 4u) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rY, (lds_t)(yb + (it * 4 + wave) * 256), 16, vo, 0, 0, 0);
    }
  };

  int t = blockIdx.x, buf = 0;
  if (t < p.ntiles) stage(t, 0);
  while (t < p.ntiles) {
    __builtin_amdgcn_s_waitcnt(0);        // this wave's DMA rows of tile t have landed ...
    __syncthreads();                      // ... and so have everyone's; all waves are also done reading the other buffer
    const int tn = t + gridDim.x;
    if (tn < p.ntiles) stage(tn, buf ^ 1);      // in flight behind this tile's MFMA stream
    const float* xl = smem + buf * (XF + YF) + lq * XP + lr;
    const float* yl = smem + buf * (XF + YF) + XF + lq * YP + lr;
    switch (wave) {
      case 0: mma_tile<PT, QT, KS, 0, NU>(acc, xl, yl); break;
      case 1: mma_tile<PT, QT, KS, 1, NU>(acc, xl, yl); break;
      case 2: mma_tile<PT, QT, KS, 2, NU>(acc, xl, yl); break;
      default: mma_tile<PT, QT, KS, 3, NU>(acc, xl, yl); break;
    }
    t = tn;
    buf ^= 1;
  }
  float* part = p.part + (int64_t)blockIdx.x * p.Cout * p.K;
  switch (wave) {
    case 0: store_partial<PT, QT, KS, 0, NU>(acc, p, part, lr, lq); break;
    case 1: store_partial<PT, QT, KS, 1, NU>(acc, p, part, lr, lq); break;
    case 2: store_partial<PT, QT, KS, 2, NU>(acc, p, part, lr, lq); break;
    default: store_partial<PT, QT, KS, 3, NU>(acc, p, part, lr, lq); break;
  }
}

}  // namespace cat_tw

namespace cat {

static int tw_tiles(int c) { return (((c + 3) & ~3) + 15) / 16; }

// which instantiation (0 = none): 1: <5,1,5>  2: <1,5,5>  3: <5,1,3>  4: <1,5,3>  5: <5,2,3>  6: <2,5,3>
static int tw_variant(const cat_conv_t* g) {
  static const int on = getenv("CAT_TWGRAD") ? atoi(getenv("CAT_TWGRAD")) : 1;
  if (!on || g->stride != 1 || g->kh != g->kw || (g->kh != 3 && g->kh != 5) || g->pad != (g->kh - 1) / 2 || g->Ho != g->H || g->Wo != g->W) return 0;
  if ((g->xcs & 3) || (g->ycs & 3) || g->Cin > 80 || g->Cout > 80) return 0;
  if ((int64_t)g->N * g->H * g->W * g->xcs * 4 >= (int64_t)2147483647 || (int64_t)g->N * g->H * g->W * g->ycs * 4 >= (int64_t)2147483647) return 0;   // 32-bit DMA offsets
  if (g->pad_mode != CAT_PAD_ZERO && (g->pad_mode != CAT_PAD_REFLECT || g->pad >= g->H || g->pad >= g->W)) return 0;
  if ((int64_t)g->N * cdiv(g->H, 8) * cdiv(g->W, 8) < 128) return 0;     // too few tiles to fill the chip: the general kernel's pixel split does better
  const int pt = tw_tiles(g->Cin), qt = tw_tiles(g->Cout);
  const int narrow = pt < qt ? pt : qt, wide = pt < qt ? qt : pt;
  if (narrow > (g->kh == 5 ? 1 : 2) || wide < 3) return 0;     // the instantiations carry 5 tiles on the wide side
  const bool xwide = pt >= qt;
  if (g->kh == 5) return xwide ? 1 : 2;
  if (narrow == 1) return xwide ? 3 : 4;
  return xwide ? 5 : 6;
}

bool twgrad_applicable(const cat_conv_t* g) { return tw_variant(g) != 0; }

int twgrad_nblk(const cat_conv_t* g) {
  const int64_t nt = (int64_t)g->N * cdiv(g->H, 8) * cdiv(g->W, 8);
  return (int)(nt < 256 ? nt : 256);
}

template <int PT, int QT, int KS>
static int tw_launch(const cat_tw::Args& a, int nblk, hipStream_t s) {
  constexpr int PH = 8 + KS - 1;
  constexpr int XIT = (PH * PH * (cat_tw::pitch_of(PT) / 4) + 255) / 256, YIT = (64 * (cat_tw::pitch_of(QT) / 4) + 255) / 256;
  const size_t lds = (size_t)2 * (XIT + YIT) * 1024 * sizeof(float);      // two buffers of whole 1 KB DMA rows
  static cat::LdsOptIn optin;
  cat::lds_optin(optin, (const void*)cat_tw::twgrad_kernel<PT, QT, KS>, (int)lds);
  cat_tw::twgrad_kernel<PT, QT, KS><<<nblk, 256, lds, s>>>(a);
  return check_launch("conv2d_twgrad");
}

// partials [nblk][Cout][taps * c4x] into ws; the caller runs the shared reduce
int twgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s) {
  cat_tw::Args a{};
  a.x = x; a.dy = dy; a.part = ws;
  a.N = g->N; a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.xcs = g->xcs; a.Cout = g->Cout; a.ycs = g->ycs; a.pad = g->pad;
  a.reflect = g->pad_mode == CAT_PAD_REFLECT;
  a.c4x = (g->Cin + 3) & ~3;
  a.c4y = (g->Cout + 3) & ~3;
  a.tiles_x = cdiv(g->W, 8);
  a.tiles_img = a.tiles_x * cdiv(g->H, 8);
  a.ntiles = g->N * a.tiles_img;
  a.K = g->kh * g->kw * a.c4x;
  const int nblk = twgrad_nblk(g);
  switch (tw_variant(g)) {
    case 1: return tw_launch<5, 1, 5>(a, nblk, s);
    case 2: return tw_launch<1, 5, 5>(a, nblk, s);
    case 3: return tw_launch<5, 1, 3>(a, nblk, s);
    case 4: return tw_launch<1, 5, 3>(a, nblk, s);
    case 5: return tw_launch<5, 2, 3>(a, nblk, s);
    case 6: return tw_launch<2, 5, 3>(a, nblk, s);
    default: set_error("twgrad: unsupported geometry"); return -22;
  }
}

}  // namespace cat
