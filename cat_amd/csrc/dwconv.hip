// Depthwise convolution (groups == C, stride 1, k in {1,3,5,...}), NHWC, gfx950.  HBM/L2-bound direct kernels: one lane owns
// a float4 of channels of one pixel, the k*k taps are re-read from L1/L2 (neighbouring lanes share them), the
// filter lives in LDS as [tap][channel] so a tap is one ds_read_b128 per lane.
// Reference: the groups=midp ConvBNReLU conv of InvertedResidualChannels (models/modules/inception_modules.py:166-173).
#include "common.h"
#include <stdlib.h>

namespace {
using cat::cdiv;

constexpr int MAXW = 16384;  // taps * cs floats of LDS filter (dynamic LDS, <= 64 KB): k=5 up to 652 channels, k=7 up to 332

struct DwArgs {
  const float* x; const float* w; const float* bias; float* y;
  int N, H, W, C, xcs, Ho, Wo, ycs, kh, kw, pad, reflect;
  int cq;        // fwd: channel quads processed per pixel (the tensor, or a channel SLICE of a wider buffer: x / y point at its first channel)
  int act;       // fwd: fused epilogue activation
  float slope;
};

__device__ __forceinline__ void load_filter(float* sw, const float* w, int C, int cs, int taps) {
  for (int i = threadIdx.x; i < taps * cs; i += 256) {
    const int t = i / cs, c = i - t * cs;
    sw[i] = c < C ? w[c * taps + t] : 0.f;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void dw_fwd_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sw[];
  const int taps = p.kh * p.kw, nq = p.cq, fcs = p.cq * 4;
  load_filter(sw, p.w, p.C, fcs, taps);
  const int64_t total = (int64_t)p.N * p.Ho * p.Wo * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
    const int ox = (int)(r % p.Wo);
    r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    const int c = cq * 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = c + e < p.C ? p.bias[c + e] : 0.f;
    }
    const float* xn = p.x + (int64_t)n * p.H * p.W * p.xcs + c;
    for (int ky = 0; ky < p.kh; ++ky) {
      int iy = oy - p.pad + ky;
      if (p.reflect) iy = cat::reflect_idx(iy, p.H);
      else if ((unsigned)iy >= (unsigned)p.H) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        int ix = ox - p.pad + kx;
        if (p.reflect) ix = cat::reflect_idx(ix, p.W);
        else if ((unsigned)ix >= (unsigned)p.W) continue;
        const f4 xv = *reinterpret_cast<const f4*>(xn + ((int64_t)iy * p.W + ix) * p.xcs);
        const f4 wv = *reinterpret_cast<const f4*>(sw + (ky * p.kw + kx) * fcs + c);
        acc += xv * wv;
      }
    }
    if (p.act != CAT_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = c + e < p.C ? cat::apply_act(acc[e], p.act, p.slope) : 0.f;
    }
    *reinterpret_cast<f4*>(p.y + (i / nq) * p.ycs + c) = acc;
  }
}

// Several depthwise convs of DIFFERENT kernel sizes over adjacent channel slices of one buffer as ONE launch (round 6: the frozen teacher's
// block, cat_amd/frozen.py -- its 1 x 1 / 3 x 3 / 5 x 5 depthwise convs with eval-mode BatchNorm folded in, plus the plain copy of the k = 1
// residual branch's hidden slice, were four launches of 10 - 32 us over the same 176-channel buffers).  Per channel quad a kernel size
// k in {1, 3, 5}; the filters sit in a 5 x 5 frame [25][cs] (k < 5: centred, the rest zero -- only the k x k window is read), a quad that is
// merely copied carries k = 1, centre weight 1, bias 0.  Lane = (pixel, quad): a pixel's quads are consecutive lanes (one coalesced row).
struct DwMultiArgs {
  const float* x; const float* w25; const float* bias; float* y;
  int N, H, W, nq, xcs, ycs, reflect, act;
  float slope;
  // runs of consecutive quads with the same kernel size: threads are dealt out run by run ([run][pixel][quad of the run]), so a wave has ONE
  // kernel size (no divergence, fully unrolled tap loops: all k * k loads of a thread in flight at once)
  int nrun;
  int run_q0[CAT_DWMULTI_MAXRUN], run_nq[CAT_DWMULTI_MAXRUN], run_k[CAT_DWMULTI_MAXRUN];
  long long run_base[CAT_DWMULTI_MAXRUN + 1];      // first thread index of the run; [nrun] = total
};

template <int K>
__device__ __forceinline__ f4 dw_multi_taps(const DwMultiArgs& p, const float* sw, const float* xn, int oy, int ox, int c, int cs, f4 acc) {
  constexpr int R = K / 2, O = 2 - R;
  int64_t offy[K];
  int offx[K];
  bool vy[K], vx[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    int iy = oy - R + t, ix = ox - R + t;
    vy[t] = p.reflect || (unsigned)iy < (unsigned)p.H;
    vx[t] = p.reflect || (unsigned)ix < (unsigned)p.W;
    iy = p.reflect ? cat::reflect_idx(iy, p.H) : (vy[t] ? iy : 0);
    ix = p.reflect ? cat::reflect_idx(ix, p.W) : (vx[t] ? ix : 0);
    offy[t] = (int64_t)iy * p.W * p.xcs;
    offx[t] = ix * p.xcs;
  }
  f4 xv[K][K];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) xv[ky][kx] = *reinterpret_cast<const f4*>(xn + offy[ky] + offx[kx]);
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const f4 wv = *reinterpret_cast<const f4*>(sw + ((O + ky) * 5 + O + kx) * cs + c);
      const float mk = (vy[ky] && vx[kx]) ? 1.f : 0.f;      // out-of-plane taps read pixel (0, 0): masked, never skipped
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv[ky][kx][e] * mk, wv[e], acc[e]);
    }
  return acc;
}

__global__ __launch_bounds__(256) void dw_multi_fwd_kernel(DwMultiArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sw[];      // [25][cs] frame, then [cs] bias
  const int cs = p.nq * 4;
  for (int i = threadIdx.x; i < 25 * p.nq; i += 256) *reinterpret_cast<f4*>(sw + i * 4) = *reinterpret_cast<const f4*>(p.w25 + i * 4);
  for (int i = threadIdx.x; i < p.nq; i += 256)
    *reinterpret_cast<f4*>(sw + 25 * cs + i * 4) = p.bias ? *reinterpret_cast<const f4*>(p.bias + i * 4) : f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int64_t total = p.run_base[p.nrun];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int r = 0;
    while (r + 1 < p.nrun && i >= p.run_base[r + 1]) ++r;
    const int64_t j = i - p.run_base[r];
    const int rq = p.run_nq[r], k = p.run_k[r];
    const int cq = p.run_q0[r] + (int)(j % rq);
    int64_t pix = j / rq;
    const int64_t opix = pix;
    const int ox = (int)(pix % p.W);
    pix /= p.W;
    const int oy = (int)(pix % p.H);
    const int n = (int)(pix / p.H);
    const int c = cq * 4;
    f4 acc = *reinterpret_cast<const f4*>(sw + 25 * cs + c);
    const float* xn = p.x + (int64_t)n * p.H * p.W * p.xcs + c;
    if (k == 1) acc = dw_multi_taps<1>(p, sw, xn, oy, ox, c, cs, acc);
    else if (k == 3) acc = dw_multi_taps<3>(p, sw, xn, oy, ox, c, cs, acc);
    else acc = dw_multi_taps<5>(p, sw, xn, oy, ox, c, cs, acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = cat::apply_act(acc[e], p.act, p.slope);
    *reinterpret_cast<f4*>(p.y + opix * p.ycs + c) = acc;
  }
}

// dxp[n,py,px,c] = sum_k dy[n, py+pe-ky, px+pe-kx, c] * w[c,ky,kx]; (Hin,Win,pe) = (H+2p,W+2p,0) for reflect, (H,W,p) for zero pad.
__global__ __launch_bounds__(256) void dw_dgrad_kernel(DwArgs p, int Hin, int Win, int pe, int dxcs) {
  extern __shared__ __attribute__((aligned(16))) float sw[];
  const int taps = p.kh * p.kw, nq = dxcs / 4;
  load_filter(sw, p.w, p.C, dxcs, taps);
  const int64_t total = (int64_t)p.N * Hin * Win * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
    const int px = (int)(r % Win);
    r /= Win;
    const int py = (int)(r % Hin);
    const int n = (int)(r / Hin);
    const int c = cq * 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* dn = p.x + (int64_t)n * p.Ho * p.Wo * p.xcs + c;  // p.x = dy here
    for (int ky = 0; ky < p.kh; ++ky) {
      const int oy = py + pe - ky;
      if ((unsigned)oy >= (unsigned)p.Ho) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ox = px + pe - kx;
        if ((unsigned)ox >= (unsigned)p.Wo) continue;
        const f4 gv = *reinterpret_cast<const f4*>(dn + ((int64_t)oy * p.Wo + ox) * p.xcs);
        const f4 wv = *reinterpret_cast<const f4*>(sw + (ky * p.kw + kx) * dxcs + c);
        acc += gv * wv;
      }
    }
    *reinterpret_cast<f4*>(p.y + i * 4) = acc;
  }
}

// dw[c][tap] partials: lanes = (pixel-lane, channel quad); each lane keeps one float4 accumulator per tap of ONE filter row
// (blockIdx.y = ky) to bound registers at kw*4; block-reduced through LDS; partial [nb][taps][cs] in ws.
// IT = index type of the pixel walk (int64_t; the int instantiation is kept switched off -- `idx32_on` -- when N*Ho*Wo < 2^31: the two div/mods per pixel are emulated at 64 bits)
template <int KW, typename IT>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwArgs p, float* __restrict__ part, int nb, int ppl) {
  __shared__ f4 red[256];
  const int nq = p.xcs / 4;
  const int tid = threadIdx.x, cq = tid % nq, pl = tid / nq;
  const int ky = blockIdx.y, b = blockIdx.x;
  const int64_t P = (int64_t)p.N * p.Ho * p.Wo;
  const int64_t per = (P + nb - 1) / nb;
  const int64_t pbeg = b * per, pend = pbeg + per < P ? pbeg + per : P;
  f4 acc[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  if (pl < ppl) {
    for (IT q = (IT)(pbeg + pl); q < (IT)pend; q += (IT)ppl) {
      const int ox = (int)(q % p.Wo);
      const IT r = q / p.Wo;
      const int oy = (int)(r % p.Ho);
      const int n = (int)(r / p.Ho);
      int iy = oy - p.pad + ky;
      if (p.reflect) iy = cat::reflect_idx(iy, p.H);
      else if ((unsigned)iy >= (unsigned)p.H) continue;
      const f4 gv = *reinterpret_cast<const f4*>(p.y + (int64_t)q * p.ycs + cq * 4);  // p.y = dy here (read only)
      const float* xr = p.x + ((int64_t)n * p.H + iy) * p.W * p.xcs + cq * 4;
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        int ix = ox - p.pad + kx;
        if (p.reflect) ix = cat::reflect_idx(ix, p.W);
        else if ((unsigned)ix >= (unsigned)p.W) continue;
        acc[kx] += gv * *reinterpret_cast<const f4*>(xr + (int64_t)ix * p.xcs);
      }
    }
  }
#pragma unroll
  for (int kx = 0; kx < KW; ++kx) {
    __syncthreads();
    red[tid] = acc[kx];
    __syncthreads();
    if (pl == 0) {
      f4 s = acc[kx];
      for (int j = 1; j < ppl; ++j) s += red[j * nq + cq];
      *reinterpret_cast<f4*>(part + ((int64_t)b * p.kh * KW + ky * KW + kx) * p.xcs + cq * 4) = s;
    }
  }
}

__global__ void dw_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int cs, int taps, int nb,
                                      int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * taps) return;
  const int c = i / taps, t = i - c * taps;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[((int64_t)b * taps + t) * cs + c];
  dw[i] = accumulate ? dw[i] + s : s;
}

int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int dw_check(const cat_conv_t* g) {
  CAT_REQUIRE(g->Cin == g->Cout && g->stride == 1, "dwconv: needs Cin == Cout, stride 1");
  CAT_REQUIRE(g->xcs % 4 == 0 && g->ycs % 4 == 0 && g->xcs >= g->Cin && g->ycs >= g->Cout, "dwconv: bad pixel stride");
  CAT_REQUIRE(g->kh * g->kw * (g->xcs > g->ycs ? g->xcs : g->ycs) <= MAXW, "dwconv: filter does not fit the LDS stage (k*k*cs <= %d)", MAXW);
  CAT_REQUIRE(g->Ho == g->H + 2 * g->pad - g->kh + 1 && g->Wo == g->W + 2 * g->pad - g->kw + 1, "dwconv: inconsistent output size");
  CAT_REQUIRE(g->pad_mode == CAT_PAD_ZERO || (g->pad < g->H && g->pad < g->W), "dwconv: reflect pad must be < input size");
  return 0;
}

DwArgs dw_args(const cat_conv_t* g) {
  DwArgs a{};
  a.N = g->N; a.H = g->H; a.W = g->W; a.C = g->Cin; a.xcs = g->xcs; a.Ho = g->Ho; a.Wo = g->Wo; a.ycs = g->ycs;
  a.kh = g->kh; a.kw = g->kw; a.pad = g->pad; a.reflect = g->pad_mode == CAT_PAD_REFLECT;
  return a;
}

struct DwWgPlan { int nb, ppl; };
DwWgPlan dw_wg_plan(const cat_conv_t* g) {
  DwWgPlan p;
  p.ppl = 256 / (g->xcs / 4);
  const int64_t P = (int64_t)g->N * g->Ho * g->Wo;
  int nb = cdiv(1024, g->kh);
  const int maxb = cdiv(P, (int64_t)p.ppl * 16);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  p.nb = nb;
  return p;
}

}  // namespace

extern "C" {

int cat_dwconv2d_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, cat_stream_t stream) {
  if (int e = dw_check(g)) return e;
  DwArgs a = dw_args(g);
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  // a channel SLICE of wider buffers: ycw (0 = whole tensor) gives the channels-with-padding this call owns; x / y point at its first one
  const int own = g->ycw > 0 ? g->ycw : g->ycs;
  CAT_REQUIRE(own % 4 == 0 && own >= g->Cin && own <= g->ycs && own <= g->xcs, "dwconv: bad channel slice (ycw=%d)", g->ycw);
  a.cq = own / 4;
  a.act = g->act;
  a.slope = g->slope;
  cat::ProfScope prof("dwconv_fwd", 2.0 * g->N * g->Ho * g->Wo * g->Cin * g->kh * g->kw, 2 * 4.0 * (double)g->N * g->Ho * g->Wo * own, stream);
  dw_fwd_kernel<<<ew_grid((int64_t)g->N * g->Ho * g->Wo * a.cq), 256, (size_t)g->kh * g->kw * own * sizeof(float), (hipStream_t)stream>>>(a);
  return cat::check_launch("dwconv2d_fwd");
}

int cat_dwconv2d_multi_fwd(const cat_dwmulti_t* g, const float* x, const float* w25, const float* bias, float* y, cat_stream_t stream) {
  CAT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->nq > 0 && g->nq <= CAT_DWMULTI_MAXQ, "dwconv multi: bad geometry (nq=%d)", g->nq);
  CAT_REQUIRE((g->xcs & 3) == 0 && (g->ycs & 3) == 0 && g->xcs >= 4 * g->nq && g->ycs >= 4 * g->nq, "dwconv multi: pixel strides");
  DwMultiArgs a{};
  a.x = x; a.w25 = w25; a.bias = bias; a.y = y;
  a.N = g->N; a.H = g->H; a.W = g->W; a.nq = g->nq; a.xcs = g->xcs; a.ycs = g->ycs; a.reflect = g->reflect; a.act = g->act; a.slope = g->slope;
  double taps = 0.0;
  const int64_t npix = (int64_t)g->N * g->H * g->W;
  a.nrun = 0;
  a.run_base[0] = 0;
  for (int q = 0; q < g->nq; ++q) {
    const int k = g->ks[q];
    CAT_REQUIRE(k == 1 || k == 3 || k == 5, "dwconv multi: kernel size %d of quad %d", k, q);
    CAT_REQUIRE(!g->reflect || (k / 2 < g->H && k / 2 < g->W), "dwconv multi: reflect padding wider than the plane");
    taps += 4.0 * k * k;
    if (a.nrun == 0 || a.run_k[a.nrun - 1] != k) {
      CAT_REQUIRE(a.nrun < CAT_DWMULTI_MAXRUN, "dwconv multi: more than %d runs of equal kernel size", CAT_DWMULTI_MAXRUN);
      a.run_q0[a.nrun] = q;
      a.run_nq[a.nrun] = 0;
      a.run_k[a.nrun] = k;
      ++a.nrun;
    }
    ++a.run_nq[a.nrun - 1];
  }
  for (int r = 0; r < a.nrun; ++r) a.run_base[r + 1] = a.run_base[r] + npix * a.run_nq[r];
  const double pix = (double)npix;
  cat::ProfScope prof("dwconv_fwd", 2.0 * pix * taps, 2 * 4.0 * pix * 4 * g->nq, stream);
  dw_multi_fwd_kernel<<<ew_grid(npix * g->nq), 256, (size_t)26 * 4 * g->nq * sizeof(float), (hipStream_t)stream>>>(a);
  return cat::check_launch("dwconv2d_multi_fwd");
}

int cat_dwconv2d_dgrad(const cat_conv_t* g, const float* dy, const float* w, float* dx, int dxcs, cat_stream_t stream) {
  if (int e = dw_check(g)) return e;
  CAT_REQUIRE(dxcs % 4 == 0 && dxcs >= g->Cin && g->kh * g->kw * dxcs <= MAXW, "dwconv dgrad: bad dxcs");
  DwArgs a = dw_args(g);
  a.x = dy; a.xcs = g->ycs; a.w = w; a.y = dx;
  cat::ProfScope prof("dwconv_dgrad", 2.0 * g->N * g->Ho * g->Wo * g->Cin * g->kh * g->kw, 2 * 4.0 * (double)g->N * g->Ho * g->Wo * g->ycs, stream);
  const bool refl = g->pad_mode == CAT_PAD_REFLECT;
  const int Hin = refl ? g->H + 2 * g->pad : g->H, Win = refl ? g->W + 2 * g->pad : g->W, pe = refl ? 0 : g->pad;
  dw_dgrad_kernel<<<ew_grid((int64_t)g->N * Hin * Win * (dxcs / 4)), 256, (size_t)g->kh * g->kw * dxcs * sizeof(float), (hipStream_t)stream>>>(a, Hin, Win, pe,
                                                                                                                                            dxcs);
  return cat::check_launch("dwconv2d_dgrad");
}

size_t cat_dwconv2d_wgrad_ws_bytes(const cat_conv_t* g) {
  return (size_t)dw_wg_plan(g).nb * g->kh * g->kw * g->xcs * sizeof(float);
}

int cat_dwconv2d_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* dw, int accumulate, void* ws, cat_stream_t stream) {
  if (int e = dw_check(g)) return e;
  CAT_REQUIRE(ws && g->xcs == g->ycs && g->xcs <= 1024, "dwconv wgrad: needs workspace and xcs == ycs <= 1024");
  DwArgs a = dw_args(g);
  a.x = x; a.y = const_cast<float*>(dy);
  const DwWgPlan pl = dw_wg_plan(g);
  hipStream_t s = (hipStream_t)stream;
  cat::ProfScope prof("dwconv_wgrad", 2.0 * g->N * g->Ho * g->Wo * g->Cin * g->kh * g->kw, 2 * 4.0 * (double)g->N * g->Ho * g->Wo * g->ycs, stream);
  dim3 grid(pl.nb, g->kh);
  constexpr int idx32_on = 0;
  const bool i32 = idx32_on && (int64_t)g->N * g->Ho * g->Wo < (int64_t)2147483647 - 65536;
#define DW_WGRAD(KW)                                                                             \
  if (i32) dw_wgrad_kernel<KW, int><<<grid, 256, 0, s>>>(a, (float*)ws, pl.nb, pl.ppl);          \
  else dw_wgrad_kernel<KW, int64_t><<<grid, 256, 0, s>>>(a, (float*)ws, pl.nb, pl.ppl);          \
  break
  switch (g->kw) {
    case 1: DW_WGRAD(1);
    case 3: DW_WGRAD(3);
    case 5: DW_WGRAD(5);
    case 7: DW_WGRAD(7);
    default: CAT_REQUIRE(false, "dwconv wgrad: kw %d unsupported", g->kw);
  }
#undef DW_WGRAD
  dw_wgrad_final_kernel<<<cdiv(g->Cin * g->kh * g->kw, 256), 256, 0, s>>>((const float*)ws, dw, g->Cin, g->xcs, g->kh * g->kw, pl.nb,
                                                                            accumulate);
  return cat::check_launch("dwconv2d_wgrad");
}

}  // extern "C"
