// Train-mode norm layers of an InvertedResidualChannels block WITHOUT their own passes over the activations (gfx950).
//
// Reference: models/modules/inception_modules.py:124-180, 230-236 -- every conv of a block is followed by norm_layer + ReLU, i.e.
// 3 HBM-bound launches (statistics, finalize, apply) per norm and 10 norms per block.  Here
//   * the producing conv leaves per-tile (sum, sum of squared deviations from the tile mean) behind (cat_tconv_fwd `stats`,
//     cat_dwm_fwd), for ALL branches of a stage in one concatenated [tile][2][channels] table;
//   * cat_tnorm_finalize merges the tiles exactly (pairwise / Chan formula: M2 = sum M2_i + sum n_i (mean_i - mean)^2), updates the
//     running statistics of every branch's norm module and emits scale = gamma * rstd, shift = beta - mean * scale for the whole
//     concatenation -- one launch per block stage instead of 3 per branch;
//   * the consumer applies scale / shift + ReLU while it stages its input tile (cat_tconv_fwd segment affine, cat_dwm_fwd), so the
//     normalised tensor is never written;  cat_affine_res_fwd is the stand-alone apply (y = act(x * scale + shift) [+ residual]) for the
//     block output and for re-materialising a hidden activation in the backward pass.
#include "common.h"

namespace {
using cat::cdiv;

constexpr int TH = 8, TW = 16;

struct FinArgs {
  cat_nslice_t sl[CAT_TNORM_MAXSLICE];
  int nsl;
};

// grid (cdiv(cs, 4), G), one wave per channel; part[((g * tiles + t) * 2 + {0,1}) * scs + c]
__global__ __launch_bounds__(256) void tnorm_finalize_kernel(const float* __restrict__ part, int scs, int tiles, int tiles_x, int Ho, int Wo,
                                                             int th, int tw, int ncls, int imgs_per_group, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, FinArgs fa, float eps, float momentum,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int cs,
                                                             int mstride) {
  // one WAVE per channel (4 channels per workgroup); tables of more than 1024 tiles (the image-resolution edge layers: 4096) take one
  // WORKGROUP per channel -- `wide` -- whose four waves split the tiles and combine through LDS
  __shared__ float xw[8];
  const bool wide = gridDim.x >= (unsigned)cs;
  const int wv = threadIdx.x >> 6;
  const int g = blockIdx.y, c = wide ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv, lane = wide ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  const int nl = wide ? 256 : 64;      // lanes that share the channel's tiles
  if (c >= cs) return;
  // which norm module owns channel c (padding channels between / behind the slices belong to none)
  int sl = -1;
  for (int k = 0; k < fa.nsl; ++k)
    if (c >= fa.sl[k].c0 && c < fa.sl[k].c0 + fa.sl[k].c) sl = k;
  const int idx = g * cs + c, midx = g * mstride + c;
  if (sl < 0) {
    if (lane == 0) {
      scale[idx] = shift[idx] = 0.f;
      if (c < mstride) mean_out[midx] = rstd_out[midx] = 0.f;
    }
    return;
  }
  const int per_img = tiles, ntile = tiles * imgs_per_group;
  const float* pg = part + (int64_t)g * ntile * 2 * scs + c;
  // every lane fetches its tiles' (sum, M2) pairs up front (independent loads: one L2 round trip instead of one per tile and pass)
  constexpr int R = 16;                  // tiles per lane kept in registers (1024 / 4096 tiles); longer tables loop
  float s = 0.f;
  float sv[R], mv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = lane + nl * r;
    const bool v = t < ntile;
    sv[r] = v ? pg[(int64_t)t * 2 * scs] : 0.f;
    mv[r] = v ? pg[(int64_t)t * 2 * scs + scs] : 0.f;
    s += sv[r];
  }
  for (int t = lane + nl * R; t < ntile; t += nl) s += pg[(int64_t)t * 2 * scs];
  s = cat::wave_sum(s);
  if (wide) {      // fixed combination order: deterministic
    if ((threadIdx.x & 63) == 0) xw[wv] = s;
    __syncthreads();
    s = (xw[0] + xw[1]) + (xw[2] + xw[3]);
  }
  const float count = (float)Ho * (float)Wo * (float)ncls * (float)imgs_per_group;
  const float mean = s / count;
  float m2 = 0.f;
  // pixels of tile t.  Planes that are whole multiples of the tile (every 64 x 64 / 128 x 128 / 256 x 256 plane of the bench) need no index
  // arithmetic: the three integer divisions per tile were ~1 500 instructions per lane -- a quarter of this latency-bound kernel's 9 us
  const bool full = Ho % th == 0 && Wo % tw == 0;
  const float nfull = (float)(th * tw);
  auto tile_n = [&](int t) {
    if (full) return nfull;
    const int ti = (t % per_img) / ncls;      // `tiles` counts entries per image: lattice tiles x classes
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    return (float)(min(th, Ho - ty * th) * min(tw, Wo - tx * tw));
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = lane + nl * r;
    if (t < ntile) {
      const float n = tile_n(t);
      const float d = sv[r] / n - mean;
      m2 += mv[r] + n * d * d;
    }
  }
  for (int t = lane + nl * R; t < ntile; t += nl) {
    const float n = tile_n(t);
    const float d = pg[(int64_t)t * 2 * scs] / n - mean;
    m2 += pg[(int64_t)t * 2 * scs + scs] + n * d * d;
  }
  m2 = cat::wave_sum(m2);
  if (wide) {
    if ((threadIdx.x & 63) == 0) xw[4 + wv] = m2;
    __syncthreads();
    m2 = (xw[4] + xw[5]) + (xw[6] + xw[7]);
  }
  if (lane != 0) return;
  float var = m2 / count;
  var = var > 0.f ? var : 0.f;
  const float rstd = rsqrtf(var + eps);
  mean_out[midx] = mean;
  rstd_out[midx] = rstd;
  const cat_nslice_t& S = fa.sl[sl];
  const int cl = c - S.c0;
  if (S.running_mean) {   // BatchNorm2d.train(): momentum update with the unbiased variance
    S.running_mean[cl] = (1.f - momentum) * S.running_mean[cl] + momentum * mean;
    const float unb = count > 1.f ? var * count / (count - 1.f) : var;
    S.running_var[cl] = (1.f - momentum) * S.running_var[cl] + momentum * unb;
    if (S.num_batches && cl == 0) *S.num_batches += 1;
  }
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  scale[idx] = ga * rstd;
  shift[idx] = be - mean * ga * rstd;
}

// SynchronizedBatchNorm over ranks for the fused units (models/modules/sync_batchnorm/batchnorm.py:103-140): the tile table of THIS rank is
// folded to [sum x | sum x^2] per channel (sum x^2 of a tile = M2 + sum^2 / n), the host all-reduces the 2 * scs floats ONCE per stage, and
// tnorm_finalize_sums_kernel applies the reference's multi-replica formula (mean = sum / size, sumvar = ssum - sum * mean,
// inv_std = clamp(sumvar / size, eps)^-1/2, running_var from the unbiased sumvar / (size - 1)) to every norm module of the stage at once.
// grid cdiv(scs, 4), one wave per channel
__global__ __launch_bounds__(256) void tnorm_sums_kernel(const float* __restrict__ part, int scs, int ntile, int per_img, int tiles_x, int Ho, int Wo,
                                                         int th, int tw, int ncls, float* __restrict__ sums) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= scs) return;
  float s = 0.f, q = 0.f;
  for (int t = lane; t < ntile; t += 64) {
    const int ti = (t % per_img) / ncls;
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const float n = (float)(min(th, Ho - ty * th) * min(tw, Wo - tx * tw));
    const float st = part[(int64_t)t * 2 * scs + c], m2 = part[(int64_t)t * 2 * scs + scs + c];
    s += st;
    q += m2 + st * st / n;
  }
  s = cat::wave_sum(s);
  q = cat::wave_sum(q);
  if (lane == 0) {
    sums[c] = s;
    sums[scs + c] = q;
  }
}

__global__ __launch_bounds__(256) void tnorm_finalize_sums_kernel(const float* __restrict__ sums, float count, int scs, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, FinArgs fa, float eps, float momentum, int clamp,
                                                                  float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ a,
                                                                  float* __restrict__ b) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= scs) return;
  int sl = -1;
  for (int k = 0; k < fa.nsl; ++k)
    if (c >= fa.sl[k].c0 && c < fa.sl[k].c0 + fa.sl[k].c) sl = k;
  if (sl < 0) {
    scale[c] = shift[c] = a[c] = b[c] = 0.f;
    return;
  }
  const float sum = sums[c], ssum = sums[scs + c];
  const float mean = sum / count;
  float sumvar = ssum - sum * mean;
  sumvar = sumvar > 0.f ? sumvar : 0.f;
  const float var = sumvar / count;
  const float rstd = clamp ? rsqrtf(fmaxf(var, eps)) : rsqrtf(var + eps);
  const cat_nslice_t& S = fa.sl[sl];
  const int cl = c - S.c0;
  if (S.running_mean) {
    const float unb = count > 1.f ? sumvar / (count - 1.f) : var;
    S.running_mean[cl] = (1.f - momentum) * S.running_mean[cl] + momentum * mean;
    S.running_var[cl] = (1.f - momentum) * S.running_var[cl] + momentum * unb;
    if (S.num_batches && cl == 0) *S.num_batches += 1;
  }
  a[c] = rstd;
  b[c] = -mean * rstd;
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  scale[c] = ga * rstd;
  shift[c] = be - mean * ga * rstd;
}

// y = act(x * scale[g][c] + shift[g][c]) (+ res): (pixel, quad) walk with 32-bit indices inside a group
__global__ __launch_bounds__(256) void affine_res_kernel(const float* __restrict__ x, int xcs, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int sstride, const float* __restrict__ res, int rcs,
                                                         float* __restrict__ y, int ycs, int Pg, int nq, int act, float slope) {
  const int g = blockIdx.y;
  const float* xg = x + (int64_t)g * Pg * xcs;
  const float* rg = res ? res + (int64_t)g * Pg * rcs : nullptr;
  float* yg = y + (int64_t)g * Pg * ycs;
  const unsigned total = (unsigned)Pg * (unsigned)nq;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned p = i / (unsigned)nq, q = i - p * (unsigned)nq;
    const f4 v = *reinterpret_cast<const f4*>(xg + (int64_t)p * xcs + q * 4);
    const f4 sc = *reinterpret_cast<const f4*>(scale + g * sstride + q * 4);
    const f4 sh = *reinterpret_cast<const f4*>(shift + g * sstride + q * 4);
    f4 o = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = cat::apply_act(o[e], act, slope);
    if (rg) o += *reinterpret_cast<const f4*>(rg + (int64_t)p * rcs + q * 4);
    *reinterpret_cast<f4*>(yg + (int64_t)p * ycs + q * 4) = o;
  }
}

// Backward of ReflectionPad2d with separate pixel strides (channel slices of wider buffers) and an optional addend:
// dx[n,y,x,:] = add[n,y,x,:] + sum of the (up to 3 x 3) padded positions that mirror onto (y, x)
__global__ __launch_bounds__(256) void reflect_fold2_kernel(const float* __restrict__ dxp, int pcs, float* __restrict__ dx, int dcs,
                                                            const float* __restrict__ add, int acs, int N, int H, int W, int nq, int pad) {
  const int64_t total = (int64_t)N * H * W * nq;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
    const int64_t pix = r;
    const int xw = (int)(r % W);
    r /= W;
    const int yh = (int)(r % H);
    const int n = (int)(r / H);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = yh + pad;
    if (yh >= 1 && yh <= pad) ys[ny++] = pad - yh;
    if (yh <= H - 2 && yh >= H - 1 - pad) ys[ny++] = pad + 2 * (H - 1) - yh;
    xs[nx++] = xw + pad;
    if (xw >= 1 && xw <= pad) xs[nx++] = pad - xw;
    if (xw <= W - 2 && xw >= W - 1 - pad) xs[nx++] = pad + 2 * (W - 1) - xw;
    f4 s = add ? *reinterpret_cast<const f4*>(add + pix * acs + cq * 4) : f4{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += *reinterpret_cast<const f4*>(dxp + (((int64_t)n * Hp + ys[a]) * Wp + xs[b]) * pcs + cq * 4);
    *reinterpret_cast<f4*>(dx + pix * dcs + cq * 4) = s;
  }
}

// ---------------------------------------------------------------------------------------------- depthwise stage of a block
// All depthwise convs of a block (k = 1 / 3 / 5 per channel quad) in one launch, reading channel slices of the first-stage buffer
// with that stage's normalise + ReLU applied while the 12 x 20 pixel patch is staged in LDS, writing the concatenated pre-norm output
// and its per-tile statistics.  One workgroup = 8 x 16 output pixels x all quads; thread = (pixel, quad parity).
struct DwmArgs {
  const float* x; const float* scale; const float* shift; const float* w; const float* bias; float* y; float* stats;
  int xcs, sstride, ycs, scs;
  int N, H, W, nq, reflect, act;
  float slope;
  int tiles_x, tiles;
  int ks[CAT_DWM_MAXQ];   // kernel size of each channel quad
};

// MAXQ = compile-time bound of p.nq (16: the training blocks; 24: frozen SPADE units with 3 x 21 -> 72 hidden channels)
template <int MAXQ>
__global__ __launch_bounds__(256) void dwm_fwd_kernel(DwmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TR = TH + 4, TC = TW + 4;
  const int cs = p.nq * 4;
  float* tile = smem;                    // [TR * TC][cs]
  float* sw = smem + TR * TC * cs;       // [25][cs] filters in a 5 x 5 frame
  float* red = sw + 25 * cs;             // [4 waves][cs]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tt = blockIdx.x, n = tt / p.tiles, t = tt - n * p.tiles;
  const int oy0 = (t / p.tiles_x) * TH, ox0 = (t % p.tiles_x) * TW;
  const int g = p.sstride ? n : 0;       // per-image statistics (InstanceNorm) or one group
  for (int i = tid; i < 25 * p.nq; i += 256) *reinterpret_cast<f4*>(sw + i * 4) = *reinterpret_cast<const f4*>(p.w + i * 4);
  const float neg = p.act == CAT_ACT_RELU ? 0.f : (p.act == CAT_ACT_LRELU ? p.slope : 1.f);
  constexpr int SIT = (TR * TC * MAXQ + 255) / 256;     // all of the thread's loads in flight before the first LDS store
  f4 xv[SIT];
  {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int i = tid + it * 256;
      xv[it] = zero;
      if (i < TR * TC * p.nq) {
        const int pix = i / p.nq, q = i - pix * p.nq;
        const int r = pix / TC, c = pix - r * TC;
        int iy = oy0 - 2 + r, ix = ox0 - 2 + c;
        bool v;
        if (p.reflect) {
          v = iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
          iy = cat::reflect_idx(iy, p.H);
          ix = cat::reflect_idx(ix, p.W);
        } else {
          v = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        }
        if (v) {
          const f4 x4 = *reinterpret_cast<const f4*>(p.x + (((int64_t)n * p.H + iy) * p.W + ix) * p.xcs + q * 4);
          const f4 sc = *reinterpret_cast<const f4*>(p.scale + g * p.sstride + q * 4);
          const f4 sh = *reinterpret_cast<const f4*>(p.shift + g * p.sstride + q * 4);
          f4 o = x4 * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * neg;
          xv[it] = o;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int i = tid + it * 256;
      if (i < TR * TC * p.nq) *reinterpret_cast<f4*>(tile + (i / p.nq) * cs + (i % p.nq) * 4) = xv[it];
    }
  }
  __syncthreads();
  const int px = tid & 127, half = tid >> 7;
  const int py = px >> 4, pxx = px & 15;
  const bool pv = oy0 + py < p.H && ox0 + pxx < p.W;
  const int cnt = min(TH, p.H - oy0) * min(TW, p.W - ox0);
  constexpr int MAXH = MAXQ / 2;
  f4 out[MAXH];
#pragma unroll
  for (int k = 0; k < MAXH; ++k) {
    const int q = half + 2 * k;
    out[k] = f4{0.f, 0.f, 0.f, 0.f};
    if (q < p.nq) {
      const int ks = p.ks[q], o = 2 - (ks >> 1);
      f4 a = p.bias ? *reinterpret_cast<const f4*>(p.bias + q * 4) : f4{0.f, 0.f, 0.f, 0.f};
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          a += *reinterpret_cast<const f4*>(tile + ((py + o + ky) * TC + pxx + o + kx) * cs + q * 4) *
               *reinterpret_cast<const f4*>(sw + ((o + ky) * 5 + o + kx) * cs + q * 4);
      out[k] = a;
      if (pv) *reinterpret_cast<f4*>(p.y + (((int64_t)n * p.H + oy0 + py) * p.W + ox0 + pxx) * p.ycs + q * 4) = a;
    }
  }
  if (!p.stats) return;
  // tile statistics: wave shuffle over its 64 pixels, the two waves of a quad parity through LDS; two passes (sum, then M2)
  float* dst = p.stats + (int64_t)tt * 2 * p.scs;
  f4 mean[MAXH];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int k = 0; k < MAXH; ++k) {
      const int q = half + 2 * k;
      if (q < p.nq) {
        f4 v = out[k];
        if (pass == 1) {
          v = v - mean[k];
          v = v * v;
        }
        if (!pv) v = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cat::wave_sum(v[e]);
        if (lane == 0) *reinterpret_cast<f4*>(red + wave * cs + q * 4) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXH; ++k) {
      const int q = half + 2 * k;
      if (q < p.nq) {
        const f4 tot = *reinterpret_cast<const f4*>(red + (2 * half) * cs + q * 4) + *reinterpret_cast<const f4*>(red + (2 * half + 1) * cs + q * 4);
        if (pass == 0) mean[k] = tot / (float)cnt;
        if ((tid & 127) == 0) *reinterpret_cast<f4*>(dst + pass * p.scs + q * 4) = tot;
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int cat_tnorm_finalize2(const float* part, int scs, int G, int N, int Ho, int Wo, int th, int tw, int ncls, const float* gamma,
                        const float* beta, int nslices, const cat_nslice_t* slices, float eps, float momentum, float* scale, float* shift,
                        float* mean, float* rstd, int mstride, cat_stream_t stream) {
  CAT_REQUIRE(G == 1 || G == N, "tnorm finalize: groups must be 1 (batch norm) or N (instance norm)");
  CAT_REQUIRE(nslices >= 1 && nslices <= CAT_TNORM_MAXSLICE && (scs & 3) == 0, "tnorm finalize: %d slices", nslices);
  FinArgs fa{};
  fa.nsl = nslices;
  for (int k = 0; k < nslices; ++k) {
    fa.sl[k] = slices[k];
    CAT_REQUIRE(slices[k].c0 >= 0 && slices[k].c > 0 && slices[k].c0 + slices[k].c <= scs, "tnorm finalize: slice %d outside the table", k);
  }
  CAT_REQUIRE(th > 0 && tw > 0 && ncls >= 1, "tnorm finalize: tile geometry");
  const int tiles_x = cdiv(Wo, tw), tiles = tiles_x * cdiv(Ho, th) * ncls;
  const int64_t ntile = (int64_t)tiles * (G == 1 ? N : 1);
  tnorm_finalize_kernel<<<dim3(ntile > 1024 ? scs : cdiv(scs, 4), G), 256, 0, (hipStream_t)stream>>>(part, scs, tiles, tiles_x, Ho, Wo, th, tw, ncls, G == 1 ? N : 1, gamma,
                                                                                  beta, fa, eps, momentum, scale, shift, mean, rstd, scs, mstride);
  return cat::check_launch("tnorm_finalize");
}

int cat_tnorm_finalize(const float* part, int scs, int G, int N, int Ho, int Wo, const float* gamma, const float* beta, int nslices,
                       const cat_nslice_t* slices, float eps, float momentum, float* scale, float* shift, float* mean, float* rstd,
                       int mstride, cat_stream_t stream) {
  return cat_tnorm_finalize2(part, scs, G, N, Ho, Wo, TH, TW, 1, gamma, beta, nslices, slices, eps, momentum, scale, shift, mean, rstd, mstride, stream);
}

int cat_tnorm_sums(const float* part, int scs, int N, int Ho, int Wo, int th, int tw, int ncls, float* sums, cat_stream_t stream) {
  CAT_REQUIRE(part && sums && (scs & 3) == 0 && N > 0 && Ho > 0 && Wo > 0 && th > 0 && tw > 0 && ncls >= 1, "tnorm sums: bad arguments");
  const int tiles_x = cdiv(Wo, tw), per_img = tiles_x * cdiv(Ho, th) * ncls;
  tnorm_sums_kernel<<<cdiv(scs, 4), 256, 0, (hipStream_t)stream>>>(part, scs, per_img * N, per_img, tiles_x, Ho, Wo, th, tw, ncls, sums);
  return cat::check_launch("tnorm_sums");
}

int cat_tnorm_finalize_sums(const float* sums, double count, int scs, const float* gamma, const float* beta, int nslices,
                            const cat_nslice_t* slices, float eps, float momentum, int clamp, float* scale, float* shift, float* a, float* b,
                            cat_stream_t stream) {
  CAT_REQUIRE(sums && scale && shift && a && b && count > 0 && (scs & 3) == 0, "tnorm finalize sums: bad arguments");
  CAT_REQUIRE(nslices >= 1 && nslices <= CAT_TNORM_MAXSLICE, "tnorm finalize sums: %d slices", nslices);
  FinArgs fa{};
  fa.nsl = nslices;
  for (int k = 0; k < nslices; ++k) {
    fa.sl[k] = slices[k];
    CAT_REQUIRE(slices[k].c0 >= 0 && slices[k].c > 0 && slices[k].c0 + slices[k].c <= scs, "tnorm finalize sums: slice %d outside the table", k);
  }
  tnorm_finalize_sums_kernel<<<cdiv(scs, 256), 256, 0, (hipStream_t)stream>>>(sums, (float)count, scs, gamma, beta, fa, eps, momentum, clamp, scale, shift,
                                                                               a, b);
  return cat::check_launch("tnorm_finalize_sums");
}

int cat_affine_res_fwd(const float* x, int xcs, const float* scale, const float* shift, int sstride, const float* res, int rcs, float* y,
                       int ycs, int G, int Pg, int C4, int act, float slope, cat_stream_t stream) {
  CAT_REQUIRE((xcs & 3) == 0 && (ycs & 3) == 0 && (C4 & 3) == 0 && C4 <= xcs && C4 <= ycs, "affine_res: channel layout");
  CAT_REQUIRE((int64_t)Pg * (C4 / 4) < (int64_t)4000000000LL, "affine_res: group too large");
  cat::ProfScope prof("affine_res", 0.0, (res ? 12.0 : 8.0) * (double)G * Pg * C4, stream);
  int64_t b = ((int64_t)Pg * (C4 / 4) + 255) / 256;
  const int gx = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
  affine_res_kernel<<<dim3(gx, G), 256, 0, (hipStream_t)stream>>>(x, xcs, scale, shift, sstride, res, rcs, y, ycs, Pg, C4 / 4, act, slope);
  return cat::check_launch("affine_res");
}

int cat_reflect_pad_bwd2(const float* dxp, int pcs, float* dx, int dcs, const float* add, int acs, int N, int H, int W, int C4, int pad,
                         cat_stream_t stream) {
  CAT_REQUIRE((pcs & 3) == 0 && (dcs & 3) == 0 && (C4 & 3) == 0 && C4 <= pcs && C4 <= dcs && pad < H && pad < W && (!add || C4 <= acs),
              "reflect_pad_bwd2: bad geometry");
  cat::ProfScope prof("reflect_pad_bwd", 0.0, 8.0 * N * H * W * C4, stream);
  const int64_t total = (int64_t)N * H * W * (C4 / 4);
  int64_t b = (total + 255) / 256;
  reflect_fold2_kernel<<<(int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)), 256, 0, (hipStream_t)stream>>>(dxp, pcs, dx, dcs, add, acs, N, H, W, C4 / 4, pad);
  return cat::check_launch("reflect_pad_bwd2");
}

int cat_dwm_fwd(const cat_dwm_t* g, const float* x, const float* scale, const float* shift, const float* w25, const float* bias, float* y,
                float* stats, cat_stream_t stream) {
  CAT_REQUIRE(g->nq >= 1 && g->nq <= CAT_DWM_MAXQ, "dwm: %d channel quads (max %d)", g->nq, CAT_DWM_MAXQ);
  CAT_REQUIRE((g->xcs & 3) == 0 && (g->ycs & 3) == 0 && g->xcs >= 4 * g->nq && g->ycs >= 4 * g->nq, "dwm: channel layout");
  CAT_REQUIRE(!g->reflect || (g->H > 2 && g->W > 2), "dwm: reflect padding wider than the plane");
  DwmArgs a{};
  a.x = x; a.scale = scale; a.shift = shift; a.w = w25; a.bias = bias; a.y = y; a.stats = stats;
  a.xcs = g->xcs; a.sstride = g->sstride; a.ycs = g->ycs; a.scs = g->scs;
  a.N = g->N; a.H = g->H; a.W = g->W; a.nq = g->nq; a.reflect = g->reflect; a.act = g->act; a.slope = g->slope;
  a.tiles_x = cdiv(g->W, TW);
  a.tiles = a.tiles_x * cdiv(g->H, TH);
  for (int q = 0; q < g->nq; ++q) {
    CAT_REQUIRE(g->ks[q] == 1 || g->ks[q] == 3 || g->ks[q] == 5, "dwm: kernel size %d", g->ks[q]);
    a.ks[q] = g->ks[q];
  }
  const int cs = g->nq * 4;
  const size_t lds = (size_t)((TH + 4) * (TW + 4) * cs + 25 * cs + 4 * cs) * sizeof(float);
  CAT_REQUIRE(lds <= 112 * 1024, "dwm: %zu bytes of LDS (max 112 KB)", lds);
  double taps = 0.0;
  for (int q = 0; q < g->nq; ++q) taps += 4.0 * g->ks[q] * g->ks[q];
  cat::ProfScope prof("dwconv_fwd", 2.0 * (double)g->N * g->H * g->W * taps, 0.0, stream);
  if (g->nq <= 16) {
    static cat::LdsOptIn optin;
    cat::lds_optin(optin, (const void*)dwm_fwd_kernel<16>, 96 * 1024);
    dwm_fwd_kernel<16><<<g->N * a.tiles, 256, lds, (hipStream_t)stream>>>(a);
  } else {
    static cat::LdsOptIn optin;
    cat::lds_optin(optin, (const void*)dwm_fwd_kernel<CAT_DWM_MAXQ>, 112 * 1024);
    dwm_fwd_kernel<CAT_DWM_MAXQ><<<g->N * a.tiles, 256, lds, (hipStream_t)stream>>>(a);
  }
  return cat::check_launch("dwm_fwd");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- per-step operand preparation
namespace {

__global__ __launch_bounds__(256) void prep_kernel(const cat_prep_job_t* __restrict__ jobs, int njobs, int accumulate) {
  // the job that owns this workgroup: block0 is ascending -- binary search (the tables of all blocks of a generator are merged into one
  // launch: ~200 jobs; a linear walk would be ~200 dependent loads in the last workgroups)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= jobs[mid].block0) lo = mid;
    else hi = mid - 1;
  }
  const int ji = lo;
  const cat_prep_job_t& J = jobs[ji];
  const int64_t e = (int64_t)(blockIdx.x - J.block0) * 256 + threadIdx.x;
  if (J.kind == 0) {   // tconv filter stream (see pack_kernel in conv_pk.hip): one float4 per thread
    const int taps = J.ks * J.ks, nfull = J.c4 >> 4, rem = (J.c4 & 15) >> 2;
    const int groups = nfull * taps + (rem ? (taps * rem + 3) / 4 : 0);
    const int nt_own0 = J.col0 >> 4, nt_own1 = (J.col0 + J.Nn + 15) >> 4;   // N tiles this conv's columns touch
    const int ntw = nt_own1 - nt_own0;
    if (e >= (int64_t)groups * ntw * 64) return;
    const int lane = (int)(e & 63);
    const int64_t gj = e >> 6;
    const int j = nt_own0 + (int)(gj % ntw);
    const int G = (int)(gj / ntw);
    int chunk, gi, nq;
    if (G < nfull * taps) { chunk = G / taps; gi = G - chunk * taps; nq = 4; }
    else { chunk = nfull; gi = G - nfull * taps; nq = rem; }
    const int lr = lane & 15, lq = lane >> 4;
    const int p = 4 * gi + lq;
    const int tap = p / nq, qd = p - tap * nq;
    const int col = j * 16 + lr, nn = col - J.col0;
    if (nn < 0 || nn >= J.Nn) return;    // another conv's column (or padding, which stays 0 from the buffer's initialisation)
    const float* w = J.srcs[0];
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (tap < taps) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = chunk * 16 + qd * 4 + t;
        if (c < J.Ck) v[t] = J.mode == 0 ? w[(int64_t)nn * J.wn + tap * J.wcs + c] : w[(int64_t)c * J.wn + (taps - 1 - tap) * J.wcs + nn];
      }
    }
    *reinterpret_cast<f4*>(J.dst + (((int64_t)G * J.nt_total + j) * 64 + lane) * 4) = v;
  } else if (J.kind == 1) {
    if (e >= J.n) return;
    float s = 0.f;
    for (int k = 0; k < J.nsrc; ++k) s += J.srcs[k][e];
    J.dst[e] = s;
  } else if (J.kind == 2) {   // depthwise filter into the 5 x 5 frame
    const int taps = J.ks * J.ks;
    if (e >= (int64_t)J.Nn * taps) return;
    const int c = (int)(e / taps), tp = (int)(e - (int64_t)c * taps);
    const int ky = tp / J.ks, kx = tp - ky * J.ks, o = 2 - (J.ks >> 1);
    J.dst[((o + ky) * 5 + o + kx) * J.cs + J.col0 + c] = J.srcs[0][e];
  } else if (J.kind == 3) {   // scatter a slice of a concatenated gradient vector to its parameter
    if (e >= J.n) return;
    float* d = const_cast<float*>(J.srcs[1]);
    const float v = J.srcs[0][e];
    d[e] = accumulate ? d[e] + v : v;
  } else {                    // kind 4: 2-D scatter, dst[r * wcs + i] (+)= src[r * wn + i], i < cs (columns of a K-concatenated weight gradient)
    if (e >= J.n) return;
    const int r = (int)(e / J.cs), i = (int)(e - (int64_t)r * J.cs);
    float* d = const_cast<float*>(J.srcs[1]) + (int64_t)r * J.wcs + i;
    const float v = J.srcs[0][(int64_t)r * J.wn + i];
    *d = accumulate ? *d + v : v;
  }
}

}  // namespace

extern "C" int cat_prep_run(const cat_prep_job_t* jobs_dev, int njobs, int total_blocks, int accumulate, cat_stream_t stream) {
  CAT_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "prep: empty job table");
  prep_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(jobs_dev, njobs, accumulate);
  return cat::check_launch("prep_run");
}

// ---------------------------------------------------------------------------------------------- depthwise stage, backward
// Input gradient and filter gradient of ALL depthwise convs of a block in one launch (+ a small final reduction): replaces, per branch,
// cat_dwconv2d_dgrad + the reflect fold + two slice copies + cat_dwconv2d_wgrad.  One workgroup = 8 x 16 pixels x all channel quads.
//   dA[i]      = sum over the padded positions j that mirror onto i (j = i, and near a border -i / 2(H-1)-i) of
//                sum_k dZ[j + p - k] * w[k]            (reflect padding; zero padding: j = i only, dZ = 0 outside the plane)
//   dW[c][k]  += sum over the tile's output pixels o of dZ[o][c] * a[reflect(o - p + k)][c]     -> per-tile partials, reduced by
//                dwm_wgrad_final_kernel straight into the parameters' gradient buffers
namespace {

struct DwmBwdArgs {
  const float* a; const float* dz; const float* w; float* da; float* part;
  int acs, zcs, dacs;
  int N, H, W, nq, reflect;
  int tiles_x, tiles;
  int ks[CAT_DWM_MAXQ];
};

__global__ __launch_bounds__(256) void dwm_bwd_kernel(DwmBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TR = TH + 4, TC = TW + 4;
  const int cs = p.nq * 4;
  float* ta = smem;                       // [TR * TC][cs]   input activation patch (reflect- or zero-padded)
  float* tz = ta + TR * TC * cs;          // [TR * TC][cs]   dZ patch (zero outside the plane)
  float* sw = tz + TR * TC * cs;          // [25][cs]
  const int tid = threadIdx.x;
  const int tt = blockIdx.x, n = tt / p.tiles, t = tt - n * p.tiles;
  const int oy0 = (t / p.tiles_x) * TH, ox0 = (t % p.tiles_x) * TW;
  for (int i = tid; i < 25 * p.nq; i += 256) *reinterpret_cast<f4*>(sw + i * 4) = *reinterpret_cast<const f4*>(p.w + i * 4);
  // staging in two phases (all global loads of the thread in flight, then the LDS stores): one workgroup per CU (89 KB of LDS), so
  // nothing else would hide a load-store-load chain
  constexpr int SIT = (TR * TC * CAT_DWM_MAXQ_BWD + 255) / 256;
  f4 zv[SIT], av[SIT];
#pragma unroll
  for (int it = 0; it < SIT; ++it) {
    const int i = tid + it * 256;
    zv[it] = av[it] = f4{0.f, 0.f, 0.f, 0.f};
    if (i < TR * TC * p.nq) {
      const int pix = i / p.nq, q = i - pix * p.nq;
      const int r = pix / TC, c = pix - r * TC;
      const int iy = oy0 - 2 + r, ix = ox0 - 2 + c;
      const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      if (in) zv[it] = *reinterpret_cast<const f4*>(p.dz + (((int64_t)n * p.H + iy) * p.W + ix) * p.zcs + q * 4);
      int ay = iy, ax = ix;
      bool av_ok = in;
      if (p.reflect) {
        av_ok = iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
        ay = cat::reflect_idx(iy, p.H);
        ax = cat::reflect_idx(ix, p.W);
      }
      if (av_ok) av[it] = *reinterpret_cast<const f4*>(p.a + (((int64_t)n * p.H + ay) * p.W + ax) * p.acs + q * 4);
    }
  }
#pragma unroll
  for (int it = 0; it < SIT; ++it) {
    const int i = tid + it * 256;
    if (i < TR * TC * p.nq) {
      const int pix = i / p.nq, q = i - pix * p.nq;
      *reinterpret_cast<f4*>(tz + pix * cs + q * 4) = zv[it];
      *reinterpret_cast<f4*>(ta + pix * cs + q * 4) = av[it];
    }
  }
  __syncthreads();
  // ---- input gradient: thread = (pixel, quad parity)
  {
    const int px = tid & 127, half = tid >> 7;
    const int py = px >> 4, pxx = px & 15;
    const int iy = oy0 + py, ix = ox0 + pxx;
    if (iy < p.H && ix < p.W) {
      // padded positions that mirror onto (iy, ix); zero padding: the pixel itself only
      int ys[3], xs[3], ny = 0, nx = 0;
      ys[ny++] = iy;
      xs[nx++] = ix;
      for (int q = half; q < p.nq; q += 2) {
        const int ks = p.ks[q], pd = ks >> 1, o = 2 - pd;
        ny = nx = 1;
        if (p.reflect) {
          if (iy >= 1 && iy <= pd) ys[ny++] = -iy;
          if (iy <= p.H - 2 && iy >= p.H - 1 - pd) ys[ny++] = 2 * (p.H - 1) - iy;
          if (ix >= 1 && ix <= pd) xs[nx++] = -ix;
          if (ix <= p.W - 2 && ix >= p.W - 1 - pd) xs[nx++] = 2 * (p.W - 1) - ix;
        }
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
          for (int b = 0; b < nx; ++b) {
            const int jy = ys[a], jx = xs[b];
            for (int ky = 0; ky < ks; ++ky) {
              const int zy = jy + pd - ky - (oy0 - 2);          // row of dZ in the patch
              if ((unsigned)zy >= (unsigned)TR) continue;
              for (int kx = 0; kx < ks; ++kx) {
                const int zx = jx + pd - kx - (ox0 - 2);
                if ((unsigned)zx >= (unsigned)TC) continue;
                acc += *reinterpret_cast<const f4*>(tz + (zy * TC + zx) * cs + q * 4) *
                       *reinterpret_cast<const f4*>(sw + ((o + ky) * 5 + o + kx) * cs + q * 4);
              }
            }
          }
        *reinterpret_cast<f4*>(p.da + (((int64_t)n * p.H + iy) * p.W + ix) * p.dacs + q * 4) = acc;
      }
    }
  }
  // ---- filter gradient partials: work item = (tap of the 5 x 5 frame, quad), summed over the tile's 128 output pixels
  float* part = p.part + (int64_t)tt * 25 * cs;
  for (int wi = tid; wi < 25 * p.nq; wi += 256) {
    const int tap = wi / p.nq, q = wi - tap * p.nq;
    const int ks = p.ks[q], o = 2 - (ks >> 1);
    const int fy = tap / 5, fx = tap - fy * 5;            // frame coordinates; the filter occupies [o, o + ks)
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (fy >= o && fy < o + ks && fx >= o && fx < o + ks) {
      for (int py = 0; py < TH; ++py) {
        if (oy0 + py >= p.H) break;
        for (int pxx = 0; pxx < TW; ++pxx) {
          if (ox0 + pxx >= p.W) break;
          acc += *reinterpret_cast<const f4*>(tz + ((py + 2) * TC + pxx + 2) * cs + q * 4) *
                 *reinterpret_cast<const f4*>(ta + ((py + fy) * TC + pxx + fx) * cs + q * 4);
        }
      }
    }
    *reinterpret_cast<f4*>(part + tap * cs + q * 4) = acc;
  }
}

struct DwmFinArgs {
  float* dst[CAT_TNORM_MAXSLICE];
  int c0[CAT_TNORM_MAXSLICE], c[CAT_TNORM_MAXSLICE], ks[CAT_TNORM_MAXSLICE];
  int nbr;
};

// dst_b[c][ky][kx] (+)= sum over tiles of part[tile][frame tap][c0_b + c]
__global__ __launch_bounds__(256) void dwm_wgrad_final_kernel(const float* __restrict__ part, int ntiles, int cs, DwmFinArgs fa, int accumulate) {
  const int b = blockIdx.y;
  const int ks = fa.ks[b], taps = ks * ks, o = 2 - (ks >> 1);
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per (channel, tap)
  if (e >= fa.c[b] * taps) return;
  const int c = e / taps, tp = e - c * taps;
  const int ky = tp / ks, kx = tp - ky * ks;
  const float* src = part + ((o + ky) * 5 + o + kx) * cs + fa.c0[b] + c;
  float s = 0.f;
  for (int t = lane; t < ntiles; t += 64) s += src[(int64_t)t * 25 * cs];
  s = cat::wave_sum(s);
  if (lane == 0) fa.dst[b][e] = accumulate ? fa.dst[b][e] + s : s;
}

}  // namespace

extern "C" {

size_t cat_dwm_bwd_ws_bytes(const cat_dwm_t* g) {
  return (size_t)g->N * cdiv(g->H, TH) * cdiv(g->W, TW) * 25 * g->nq * 4 * sizeof(float);
}

int cat_dwm_bwd(const cat_dwm_t* g, const float* a, const float* dz, const float* w25, float* da, int dacs, int nbranch, const int* c0,
                const int* c, const int* ks, float* const* dw, int accumulate, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(g->nq >= 1 && g->nq <= CAT_DWM_MAXQ_BWD && nbranch >= 1 && nbranch <= CAT_TNORM_MAXSLICE && ws, "dwm bwd: bad arguments");
  CAT_REQUIRE((g->xcs & 3) == 0 && (g->ycs & 3) == 0 && (dacs & 3) == 0 && g->xcs >= 4 * g->nq && g->ycs >= 4 * g->nq && dacs >= 4 * g->nq,
              "dwm bwd: channel layout");
  DwmBwdArgs p{};
  p.a = a; p.dz = dz; p.w = w25; p.da = da; p.part = (float*)ws;
  p.acs = g->xcs; p.zcs = g->ycs; p.dacs = dacs;
  p.N = g->N; p.H = g->H; p.W = g->W; p.nq = g->nq; p.reflect = g->reflect;
  p.tiles_x = cdiv(g->W, TW);
  p.tiles = p.tiles_x * cdiv(g->H, TH);
  double taps = 0.0;
  for (int q = 0; q < g->nq; ++q) {
    p.ks[q] = g->ks[q];
    taps += 4.0 * g->ks[q] * g->ks[q];
  }
  const int cs = g->nq * 4, ntiles = g->N * p.tiles;
  const size_t lds = (size_t)(2 * (TH + 4) * (TW + 4) * cs + 25 * cs) * sizeof(float);
  CAT_REQUIRE(lds <= 144 * 1024, "dwm bwd: %zu bytes of LDS (max 144 KB)", lds);
  static cat::LdsOptIn optin;
  cat::lds_optin(optin, (const void*)dwm_bwd_kernel, 144 * 1024);
  hipStream_t s = (hipStream_t)stream;
  cat::ProfScope prof("dwconv_bwd", 4.0 * (double)g->N * g->H * g->W * taps, 0.0, stream);
  dwm_bwd_kernel<<<ntiles, 256, lds, s>>>(p);
  if (int e = cat::check_launch("dwm_bwd")) return e;
  DwmFinArgs fa{};
  fa.nbr = nbranch;
  int maxe = 0;
  for (int b = 0; b < nbranch; ++b) {
    fa.dst[b] = dw[b]; fa.c0[b] = c0[b]; fa.c[b] = c[b]; fa.ks[b] = ks[b];
    maxe = c[b] * ks[b] * ks[b] > maxe ? c[b] * ks[b] * ks[b] : maxe;
  }
  dwm_wgrad_final_kernel<<<dim3(cdiv(maxe, 4), nbranch), 256, 0, s>>>((const float*)ws, ntiles, cs, fa, accumulate);
  return cat::check_launch("dwm_wgrad_final");
}

}  // extern "C"
