// GauGAN / SPADE distillation path, gfx950.  Everything here is HBM-bound streaming work on NHWC
// activations (pixel stride cs = round_up(C,4) floats, float4 per lane, a wave touches whole pixels):
//   * split-phase batch norm  -- local [sum x, sum x^2] -> (RCCL all-reduce between the calls, done by the host) -> mean / inv_std
//     -> apply; backward likewise with [sum g, sum g*xhat].  One code path serves BatchNorm on one GPU and
//     SynchronizedBatchNorm over ranks (models/modules/sync_batchnorm/batchnorm.py:68-140).
//   * SPADE modulation  y = act(xhat * (1 + gamma) + beta)  fused with the activation behind it, and its two-pass backward
//     (models/modules/inception_modules.py:746-762, :549-553)
//   * nearest interpolation / x2 upsampling, 3x3 s2 average pooling without pad count, 2x2 max pooling, one-hot + instance
//     edges, spectral-norm power iteration (one step per forward, torch.nn.utils.spectral_norm semantics).
#include "common.h"

namespace {
using cat::cdiv;

int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// ------------------------------------------------------------------------------------------ per-channel column statistics
struct ColPlan {
  int nq, nz, zq, ppl, nb;
};

ColPlan col_plan(int64_t M, int cs) {
  ColPlan p;
  p.nq = cs / 4;
  p.nz = cdiv(p.nq, 256);
  p.zq = cdiv(p.nq, p.nz);
  p.ppl = 256 / p.zq;
  int nb = cdiv(2048, p.nz);
  const int maxb = cdiv(M, (int64_t)p.ppl * 8);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  p.nb = nb;
  return p;
}

struct ColArgs {
  const float* x;       // activation [M][cs]
  const float* dy;      // MODE 1/2: gradient w.r.t. the activated output
  const float* y;       // MODE 2: activated output (activation mask)
  const float* gamma;   // MODE 1: affine weight (may be null)
  const float* beta;    // MODE 1
  const float* a;       // MODE 1/2: xhat = x * a[c] + b[c]   (a = inv_std, b = -mean * inv_std), length cs, zero on padding
  const float* b;
  const float* gb;      // MODE 2: [M][gcs] gamma | beta maps
  float* dgb;           // MODE 2: [M][gcs]
  float* dxh;           // MODE 2: [M][cs]  g * (1 + gamma)
  float* part;          // [nb][2][cs]
  int64_t M;
  int C, cs, gcs, zq, ppl, nb, act;
  float slope;
};

// MODE 0: sum(x - x0), sum((x - x0)^2)  (x0 = first pixel: keeps the fp32 sums well conditioned)
// MODE 1: sum(g), sum(g * xhat), g = dy * act'(gamma * xhat + beta)
// MODE 2: SPADE: g = dy * act'(y); writes dgamma = g * xhat, dbeta = g, dxh = g * (1 + gamma); sums dxh, dxh * xhat
template <int MODE>
__global__ __launch_bounds__(256) void col_stats_kernel(const ColArgs A) {
  __shared__ f4 red[2][256];
  const int b = blockIdx.x, z = blockIdx.y;
  const int tid = threadIdx.x;
  const int cq_l = tid % A.zq, pl = tid / A.zq;
  const int cq = z * A.zq + cq_l;
  const int cs = A.cs, C = A.C;
  const bool active = pl < A.ppl && cq * 4 < cs;
  const int64_t per = (A.M + A.nb - 1) / A.nb;
  const int64_t pbeg = b * per, pend = min(A.M, pbeg + per);
  f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int c = cq * 4;
    if (MODE == 0) {
      const f4 sh = *reinterpret_cast<const f4*>(A.x + c);
      for (int64_t p = pbeg + pl; p < pend; p += A.ppl) {
        const f4 v = *reinterpret_cast<const f4*>(A.x + p * cs + c) - sh;
        s0 += v;
        s1 += v * v;
      }
    } else {
      const f4 av = *reinterpret_cast<const f4*>(A.a + c), bv = *reinterpret_cast<const f4*>(A.b + c);
      if (MODE == 1) {
        f4 ga, be;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool cv = c + e < C;
          ga[e] = cv ? (A.gamma ? A.gamma[c + e] : 1.f) : 0.f;
          be[e] = cv ? (A.beta ? A.beta[c + e] : 0.f) : 0.f;
        }
        for (int64_t p = pbeg + pl; p < pend; p += A.ppl) {
          const f4 xh = *reinterpret_cast<const f4*>(A.x + p * cs + c) * av + bv;
          f4 gv = *reinterpret_cast<const f4*>(A.dy + p * cs + c);
          if (A.act != CAT_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              gv[e] *= cat::act_grad_from_out(cat::apply_act(ga[e] * xh[e] + be[e], A.act, A.slope), A.act, A.slope);
          }
          s0 += gv;
          s1 += gv * xh;
        }
      } else {
        const bool vec = (C % 4) == 0;
        for (int64_t p = pbeg + pl; p < pend; p += A.ppl) {
          const f4 xh = *reinterpret_cast<const f4*>(A.x + p * cs + c) * av + bv;
          f4 gv = *reinterpret_cast<const f4*>(A.dy + p * cs + c);
          if (A.act != CAT_ACT_NONE) {
            const f4 yv = *reinterpret_cast<const f4*>(A.y + p * cs + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] *= cat::act_grad_from_out(yv[e], A.act, A.slope);
          }
          f4 gm;
          const float* gp = A.gb + p * A.gcs;
          float* dg = A.dgb + p * A.gcs;
          if (vec) {
            gm = *reinterpret_cast<const f4*>(gp + c);
            *reinterpret_cast<f4*>(dg + c) = gv * xh;
            *reinterpret_cast<f4*>(dg + C + c) = gv;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool cv = c + e < C;
              gm[e] = cv ? gp[c + e] : 0.f;
              if (cv) {
                dg[c + e] = gv[e] * xh[e];
                dg[C + c + e] = gv[e];
              }
            }
          }
          const f4 dxh = gv * (gm + 1.f);
          *reinterpret_cast<f4*>(A.dxh + p * cs + c) = dxh;
          s0 += dxh;
          s1 += dxh * xh;
        }
      }
    }
  }
  red[0][tid] = s0;
  red[1][tid] = s1;
  __syncthreads();
  if (pl == 0 && cq * 4 < cs) {
    for (int j = 1; j < A.ppl; ++j) {
      s0 += red[0][j * A.zq + cq_l];
      s1 += red[1][j * A.zq + cq_l];
    }
    float* dst = A.part + ((int64_t)b * 2) * cs + cq * 4;
    *reinterpret_cast<f4*>(dst) = s0;
    *reinterpret_cast<f4*>(dst + cs) = s1;
  }
}

// sums[0][c], sums[1][c] = column sums of the nb partials; block = 16 channels x 16 partial lanes.
// shifted != 0: the partials are sums about x0 = x[c] (first pixel) and are converted to raw sum x, sum x^2.
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ part, const float* __restrict__ x,
                                                         float* __restrict__ sums, int C, int cs, int nb, int shifted, float M) {
  __shared__ float red[2][256];
  const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    for (int b = j; b < nb; b += 16) {
      const float* src = part + ((int64_t)b * 2) * cs + c;
      s0 += src[0];
      s1 += src[cs];
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 8; o > 0; o >>= 1) {
    if (j < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o * 16];
      red[1][threadIdx.x] += red[1][threadIdx.x + o * 16];
    }
    __syncthreads();
  }
  if (j != 0 || c >= cs) return;
  s0 = c < C ? red[0][cl] : 0.f;
  s1 = c < C ? red[1][cl] : 0.f;
  if (shifted && c < C) {
    const float x0 = x[c];
    s1 = s1 + 2.f * x0 * s0 + M * x0 * x0;
    s0 = s0 + M * x0;
  }
  sums[c] = s0;
  sums[cs + c] = s1;
}

// mean / inv_std from the (all-reduced) raw sums, running-stat update and the affine map y = x * scale + shift.
// clamp != 0: inv_std = max(var, eps)^-1/2 (sync_batchnorm/batchnorm.py:140, the multi-replica path);
// clamp == 0: (var + eps)^-1/2 (F.batch_norm, the single-device path, batchnorm.py:69-72).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, float count, int C, int cs, float eps, int clamp,
                                                          float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o, float* __restrict__ rm,
                                                          float* __restrict__ rv, float* __restrict__ a, float* __restrict__ b,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cs) return;
  if (c >= C) {
    a[c] = b[c] = 0.f;
    if (scale) scale[c] = shift[c] = 0.f;
    return;
  }
  const float sum = sums[c], ssum = sums[cs + c];
  const float mean = sum / count;
  float sumvar = ssum - sum * mean;
  sumvar = sumvar > 0.f ? sumvar : 0.f;
  const float var = sumvar / count;
  const float rstd = clamp ? rsqrtf(fmaxf(var, eps)) : rsqrtf(var + eps);
  mean_o[c] = mean;
  rstd_o[c] = rstd;
  if (rm) {
    const float unb = count > 1.f ? sumvar / (count - 1.f) : var;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * unb;
  }
  a[c] = rstd;
  b[c] = -mean * rstd;
  if (scale) {
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    scale[c] = ga * rstd;
    shift[c] = be - mean * ga * rstd;
  }
}

// dx = gamma * inv_std * (g - sum(g)/count - xhat * sum(g*xhat)/count); dgamma / dbeta from the LOCAL sums
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ sums, float inv_count, float* __restrict__ dx,
                                                           int64_t nquads, int nq, int C, int cs, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const f4 av = *reinterpret_cast<const f4*>(a + c), bv = *reinterpret_cast<const f4*>(b + c);
    const f4 m1 = *reinterpret_cast<const f4*>(sums + c) * inv_count, m2 = *reinterpret_cast<const f4*>(sums + cs + c) * inv_count;
    const f4 xh = *reinterpret_cast<const f4*>(x + i * 4) * av + bv;
    f4 gv = *reinterpret_cast<const f4*>(dy + i * 4);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool cv = c + e < C;
      const float ga = cv ? (gamma ? gamma[c + e] : 1.f) : 0.f;
      if (act != CAT_ACT_NONE) {
        const float be = cv ? (beta ? beta[c + e] : 0.f) : 0.f;
        gv[e] *= cat::act_grad_from_out(cat::apply_act(ga * xh[e] + be, act, slope), act, slope);
      }
      o[e] = ga * av[e] * (gv[e] - m1[e] - xh[e] * m2[e]);
    }
    *reinterpret_cast<f4*>(dx + i * 4) = o;
  }
}

__global__ void bn_param_grad_kernel(const float* __restrict__ local_sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                     int cs, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + local_sums[c] : local_sums[c];
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + local_sums[cs + c] : local_sums[cs + c];
}

// ------------------------------------------------------------------------------------------ SPADE modulation
__global__ __launch_bounds__(256) void spade_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                        const float* __restrict__ b, const float* __restrict__ gb, float* __restrict__ y,
                                                        int64_t nquads, int nq, int C, int cs, int gcs, int act, float slope) {
  const bool vec = (C % 4) == 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const int64_t p = i / nq;
    const f4 xh = *reinterpret_cast<const f4*>(x + i * 4) * *reinterpret_cast<const f4*>(a + c) + *reinterpret_cast<const f4*>(b + c);
    const float* gp = gb + p * gcs;
    f4 gm, bt;
    if (vec) {
      gm = *reinterpret_cast<const f4*>(gp + c);
      bt = *reinterpret_cast<const f4*>(gp + C + c);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cv = c + e < C;
        gm[e] = cv ? gp[c + e] : 0.f;
        bt[e] = cv ? gp[C + c + e] : 0.f;
      }
    }
    f4 o = xh * (gm + 1.f) + bt;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = c + e < C ? cat::apply_act(o[e], act, slope) : 0.f;
    *reinterpret_cast<f4*>(y + i * 4) = o;
  }
}

// dx = inv_std * (dxh - sum(dxh)/count - xhat * sum(dxh*xhat)/count), in place on the dxh buffer
__global__ __launch_bounds__(256) void spade_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ b, const float* __restrict__ sums, float inv_count,
                                                              float* __restrict__ dx, int64_t nquads, int nq, int cs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const f4 av = *reinterpret_cast<const f4*>(a + c);
    const f4 xh = *reinterpret_cast<const f4*>(x + i * 4) * av + *reinterpret_cast<const f4*>(b + c);
    const f4 m1 = *reinterpret_cast<const f4*>(sums + c) * inv_count, m2 = *reinterpret_cast<const f4*>(sums + cs + c) * inv_count;
    const f4 d = *reinterpret_cast<const f4*>(dx + i * 4);
    *reinterpret_cast<f4*>(dx + i * 4) = av * (d - m1 - xh * m2);
  }
}

// ------------------------------------------------------------------------------------------ resampling / pooling
__global__ __launch_bounds__(256) void interp_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int Hi,
                                                             int Wi, int Ho, int Wo, int nq, int xcs, int ycs, float sh, float sw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int hi = min((int)floorf(ho * sh), Hi - 1), wi = min((int)floorf(wo * sw), Wi - 1);
    *reinterpret_cast<f4*>(y + (((int64_t)n * Ho + ho) * Wo + wo) * ycs + q * 4) =
        *reinterpret_cast<const f4*>(x + (((int64_t)n * Hi + hi) * Wi + wi) * xcs + q * 4);
  }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int Hi,
                                                           int Wi, int f, int nq, int cs) {
  const int Wo = Wi * f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int wi = (int)(p % Wi);
    p /= Wi;  // p = n * Hi + hi
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int aa = 0; aa < f; ++aa)
      for (int bb = 0; bb < f; ++bb) s += *reinterpret_cast<const f4*>(dy + ((p * f + aa) * Wo + wi * f + bb) * cs + q * 4);
    *reinterpret_cast<f4*>(dx + i * 4) = s;
  }
}

// F.avg_pool2d(kernel 3, stride 2, padding 1, count_include_pad=False)  (discriminators.py:213-218)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int H, int W,
                                                          int Ho, int Wo, int nq, int cs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int h0 = max(2 * ho - 1, 0), h1 = min(2 * ho + 1, H - 1), w0 = max(2 * wo - 1, 0), w1 = min(2 * wo + 1, W - 1);
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int h = h0; h <= h1; ++h)
      for (int w = w0; w <= w1; ++w) s += *reinterpret_cast<const f4*>(x + (((int64_t)n * H + h) * W + w) * cs + q * 4);
    *reinterpret_cast<f4*>(y + i * 4) = s / (float)((h1 - h0 + 1) * (w1 - w0 + 1));
  }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int H, int W,
                                                          int Ho, int Wo, int nq, int cs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int ho = h >> 1; ho <= min((h + 1) >> 1, Ho - 1); ++ho) {
      const int ch = min(2 * ho + 1, H - 1) - max(2 * ho - 1, 0) + 1;
      for (int wo = w >> 1; wo <= min((w + 1) >> 1, Wo - 1); ++wo) {
        const int cw = min(2 * wo + 1, W - 1) - max(2 * wo - 1, 0) + 1;
        s += *reinterpret_cast<const f4*>(dy + (((int64_t)n * Ho + ho) * Wo + wo) * cs + q * 4) / (float)(ch * cw);
      }
    }
    *reinterpret_cast<f4*>(dx + i * 4) = s;
  }
}

// nn.MaxPool2d(2, 2) (VGG19 features, models/modules/loss.py:151-186).  Backward re-derives the arg-max from x with the
// forward's rule (first maximum in row-major window order wins, as ATen's max_pool2d_with_indices).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int H, int W,
                                                          int Ho, int Wo, int nq, int cs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float* base = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * cs + q * 4;
    f4 m = *reinterpret_cast<const f4*>(base);
    const f4 v1 = *reinterpret_cast<const f4*>(base + cs), v2 = *reinterpret_cast<const f4*>(base + (int64_t)W * cs),
             v3 = *reinterpret_cast<const f4*>(base + (int64_t)W * cs + cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(m[e], v1[e]), fmaxf(v2[e], v3[e]));
    *reinterpret_cast<f4*>(y + i * 4) = m;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                          int64_t total, int H, int W, int Ho, int Wo, int nq, int cs) {
  // one lane per OUTPUT quad: scatters dy to the arg-max of its own 2x2 window (windows do not overlap) and zeros the rest
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t p = i / nq;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int64_t off = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * cs + q * 4;
    const int64_t o1 = off + cs, o2 = off + (int64_t)W * cs, o3 = o2 + cs;
    const f4 v0 = *reinterpret_cast<const f4*>(x + off), v1 = *reinterpret_cast<const f4*>(x + o1),
             v2 = *reinterpret_cast<const f4*>(x + o2), v3 = *reinterpret_cast<const f4*>(x + o3);
    const f4 g = *reinterpret_cast<const f4*>(dy + i * 4);
    f4 d0, d1, d2, d3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int k = 0;
      float m = v0[e];
      if (v1[e] > m) { m = v1[e]; k = 1; }
      if (v2[e] > m) { m = v2[e]; k = 2; }
      if (v3[e] > m) { m = v3[e]; k = 3; }
      d0[e] = k == 0 ? g[e] : 0.f;
      d1[e] = k == 1 ? g[e] : 0.f;
      d2[e] = k == 2 ? g[e] : 0.f;
      d3[e] = k == 3 ? g[e] : 0.f;
    }
    *reinterpret_cast<f4*>(dx + off) = d0;
    *reinterpret_cast<f4*>(dx + o1) = d1;
    *reinterpret_cast<f4*>(dx + o2) = d2;
    *reinterpret_cast<f4*>(dx + o3) = d3;
  }
}

// SPADEModel.preprocess_input / get_edges (models/spade_model.py:142-179): one-hot of the label map + 4-neighbour instance edges
__global__ __launch_bounds__(256) void onehot_edges_kernel(const int* __restrict__ label, const int* __restrict__ inst, float* __restrict__ y,
                                                           int64_t total, int H, int W, int nc, int nq, int cs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    const int64_t p = i / nq;
    const int lab = label[p];
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (q * 4 + e == lab && lab < nc) ? 1.f : 0.f;
    if (inst && nc >= q * 4 && nc < q * 4 + 4) {
      const int w = (int)(p % W), h = (int)((p / W) % H);
      const int t = inst[p];
      bool edge = false;
      if (w > 0) edge |= inst[p - 1] != t;
      if (w < W - 1) edge |= inst[p + 1] != t;
      if (h > 0) edge |= inst[p - W] != t;
      if (h < H - 1) edge |= inst[p + W] != t;
      o[nc - q * 4] = edge ? 1.f : 0.f;
    }
    *reinterpret_cast<f4*>(y + i * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------ spectral normalisation
// Weight matrix W_mat = weight.view(O, -1) in TORCH order (column j = (i, kh, kw)); storage is [O][taps][wcs].
// v is kept in torch order (state_dict 'weight_v'), vp is its copy in storage order (zero on padding lanes).
// t[z][j] = sum over the z-th slice of rows o of W[o][j] * u[o]   (grid (cdiv(Kp,256), SN_OSPLIT); sn_norm_v_kernel adds the slices)
constexpr int SN_OSPLIT = 16;
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ t, int O,
                                                      int Kp) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Kp) return;
  const int per = (O + SN_OSPLIT - 1) / SN_OSPLIT;
  const int o0 = blockIdx.y * per, o1 = min(O, o0 + per);
  float s = 0.f;
  for (int o = o0; o < o1; ++o) s += w[(int64_t)o * Kp + j] * u[o];
  t[(int64_t)blockIdx.y * Kp + j] = s;
}

__device__ float block_sum(float v, float* red) {
  v = cat::wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// v = t / max(||t||, eps), written both in storage order (vp) and torch order (v)
__global__ __launch_bounds__(1024) void sn_norm_v_kernel(float* __restrict__ t, float* __restrict__ vp, float* __restrict__ v, int I,
                                                         int taps, int wcs, float eps) {
  __shared__ float red[16];
  const int Kp = taps * wcs;
  float s = 0.f;
  for (int j = threadIdx.x; j < Kp; j += blockDim.x) {
    float a = 0.f;
    for (int z = 0; z < SN_OSPLIT; ++z) a += t[(int64_t)z * Kp + j];
    t[j] = a;      // slice 0 now holds the full W^T u (each j is owned by one thread)
    s += (j % wcs) < I ? a * a : 0.f;
  }
  const float nrm = sqrtf(block_sum(s, red));
  const float inv = 1.f / fmaxf(nrm, eps);
  for (int j = threadIdx.x; j < Kp; j += blockDim.x) {
    const int i = j % wcs, tp = j / wcs;
    const float val = i < I ? t[j] * inv : 0.f;
    vp[j] = val;
    if (i < I) v[i * taps + tp] = val;
  }
}

__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ vp, float* __restrict__ s, int Kp) {
  __shared__ float red[4];
  const int o = blockIdx.x;
  float acc = 0.f;
  for (int j = threadIdx.x; j < Kp; j += 256) acc += w[(int64_t)o * Kp + j] * vp[j];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) s[o] = acc;
}

// u = s / max(||s||, eps); sigma = u . s
__global__ __launch_bounds__(1024) void sn_norm_u_kernel(const float* __restrict__ s, float* __restrict__ u, float* __restrict__ sigma, int O,
                                                         float eps) {
  __shared__ float red[16];
  float a = 0.f;
  for (int o = threadIdx.x; o < O; o += blockDim.x) a += s[o] * s[o];
  const float n2 = block_sum(a, red);
  const float inv = 1.f / fmaxf(sqrtf(n2), eps);
  for (int o = threadIdx.x; o < O; o += blockDim.x) u[o] = s[o] * inv;
  if (threadIdx.x == 0) sigma[0] = n2 * inv;
}

// sigma = u . (W v) without a power iteration (eval mode)
__global__ __launch_bounds__(1024) void sn_sigma_kernel(const float* __restrict__ s, const float* __restrict__ u, float* __restrict__ sigma,
                                                        int O) {
  __shared__ float red[16];
  float a = 0.f;
  for (int o = threadIdx.x; o < O; o += blockDim.x) a += s[o] * u[o];
  a = block_sum(a, red);
  if (threadIdx.x == 0) sigma[0] = a;
}

__global__ __launch_bounds__(256) void sn_vp_kernel(const float* __restrict__ v, float* __restrict__ vp, int I, int taps, int wcs) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= taps * wcs) return;
  const int i = j % wcs, tp = j / wcs;
  vp[j] = i < I ? v[i * taps + tp] : 0.f;
}

__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ out,
                                                       int64_t n4) {
  const float inv = 1.f / sigma[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    *reinterpret_cast<f4*>(out + i * 4) = *reinterpret_cast<const f4*>(w + i * 4) * inv;
}

// part[b] = partial of sum(gw * w_sn)
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ gw, const float* __restrict__ wsn, float* __restrict__ part,
                                                     int64_t n) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += gw[i] * wsn[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// d weight_orig (+)= (gw - dot * u v^T) / sigma
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ gw, const float* __restrict__ part, int npart,
                                                     const float* __restrict__ u, const float* __restrict__ vp, const float* __restrict__ sigma,
                                                     float* __restrict__ dw, int64_t n, int Kp, int accumulate) {
  float dot = 0.f;
  for (int i = 0; i < npart; ++i) dot += part[i];
  const float inv = 1.f / sigma[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int o = (int)(i / Kp), j = (int)(i % Kp);
    const float g = (gw[i] - dot * u[o] * vp[j]) * inv;
    dw[i] = accumulate ? dw[i] + g : g;
  }
}

}  // namespace

extern "C" {

size_t cat_bn_ws_bytes(int64_t M, int cs) { return (size_t)col_plan(M, cs).nb * 2 * cs * sizeof(float); }

static int launch_col(int mode, ColArgs& A, const ColPlan& p, hipStream_t s) {
  A.zq = p.zq;
  A.ppl = p.ppl;
  A.nb = p.nb;
  const dim3 grid(p.nb, p.nz);
  if (mode == 0) col_stats_kernel<0><<<grid, 256, 0, s>>>(A);
  else if (mode == 1) col_stats_kernel<1><<<grid, 256, 0, s>>>(A);
  else col_stats_kernel<2><<<grid, 256, 0, s>>>(A);
  return 0;
}

int cat_bn_stats_fwd(const float* x, int64_t M, int C, int cs, float* sums, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && M > 0 && ws && sums, "bn_stats_fwd: bad arguments");
  cat::ProfScope prof("bn_stats", 0.0, 4.0 * (double)M * cs, stream);
  hipStream_t s = (hipStream_t)stream;
  const ColPlan p = col_plan(M, cs);
  ColArgs A = {};
  A.x = x;
  A.part = (float*)ws;
  A.M = M;
  A.C = C;
  A.cs = cs;
  launch_col(0, A, p, s);
  col_reduce_kernel<<<cdiv(cs, 16), 256, 0, s>>>((const float*)ws, x, sums, C, cs, p.nb, 1, (float)M);
  return cat::check_launch("bn_stats_fwd");
}

int cat_bn_finalize(const float* sums, double count, int C, int cs, float eps, int clamp, float momentum, const float* gamma,
                    const float* beta, float* mean, float* rstd, float* running_mean, float* running_var, float* a, float* b, float* scale,
                    float* shift, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && count > 0 && mean && rstd && a && b, "bn_finalize: bad arguments");
  bn_finalize_kernel<<<cdiv(cs, 256), 256, 0, (hipStream_t)stream>>>(sums, (float)count, C, cs, eps, clamp, momentum, gamma, beta, mean, rstd,
                                                                     running_mean, running_var, a, b, scale, shift);
  return cat::check_launch("bn_finalize");
}

int cat_bn_stats_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* a, const float* b, int64_t M, int C,
                     int cs, int act, float slope, float* sums, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && M > 0 && ws && sums, "bn_stats_bwd: bad arguments");
  cat::ProfScope prof("bn_stats", 0.0, 8.0 * (double)M * cs, stream);
  hipStream_t s = (hipStream_t)stream;
  const ColPlan p = col_plan(M, cs);
  ColArgs A = {};
  A.x = x;
  A.dy = dy;
  A.gamma = gamma;
  A.beta = beta;
  A.a = a;
  A.b = b;
  A.part = (float*)ws;
  A.M = M;
  A.C = C;
  A.cs = cs;
  A.act = act;
  A.slope = slope;
  launch_col(1, A, p, s);
  col_reduce_kernel<<<cdiv(cs, 16), 256, 0, s>>>((const float*)ws, nullptr, sums, C, cs, p.nb, 0, 0.f);
  return cat::check_launch("bn_stats_bwd");
}

int cat_bn_apply_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* a, const float* b,
                     const float* sums, double count, const float* local_sums, float* dx, float* dgamma, float* dbeta, int accumulate,
                     int64_t M, int C, int cs, int act, float slope, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && M > 0 && count > 0, "bn_apply_bwd: bad arguments");
  cat::ProfScope prof("bn_apply_bwd", 0.0, 12.0 * (double)M * cs, stream);
  hipStream_t s = (hipStream_t)stream;
  if (dgamma || dbeta) bn_param_grad_kernel<<<cdiv(C, 256), 256, 0, s>>>(local_sums, dgamma, dbeta, C, cs, accumulate);
  if (dx) {
    const int64_t nquads = M * (cs / 4);
    bn_bwd_apply_kernel<<<ew_grid(nquads), 256, 0, s>>>(x, dy, gamma, beta, a, b, sums, (float)(1.0 / count), dx, nquads, cs / 4, C, cs, act,
                                                         slope);
  }
  return cat::check_launch("bn_apply_bwd");
}

int cat_spade_fwd(const float* x, const float* a, const float* b, const float* gb, float* y, int64_t M, int C, int cs, int gcs, int act,
                  float slope, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && gcs % 4 == 0 && gcs >= 2 * C && M > 0, "spade_fwd: bad geometry");
  cat::ProfScope prof("spade_fwd", 0.0, 4.0 * (double)M * (2 * cs + gcs), stream);
  const int64_t nquads = M * (cs / 4);
  spade_fwd_kernel<<<ew_grid(nquads), 256, 0, (hipStream_t)stream>>>(x, a, b, gb, y, nquads, cs / 4, C, cs, gcs, act, slope);
  return cat::check_launch("spade_fwd");
}

int cat_spade_bwd_stats(const float* x, const float* a, const float* b, const float* gb, const float* y, const float* dy, float* dgb,
                        float* dxh, float* sums, int64_t M, int C, int cs, int gcs, int act, float slope, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && gcs % 4 == 0 && gcs >= 2 * C && M > 0 && ws, "spade_bwd_stats: bad geometry");
  cat::ProfScope prof("spade_bwd", 0.0, 4.0 * (double)M * (5 * cs + 2 * gcs), stream);
  hipStream_t s = (hipStream_t)stream;
  const ColPlan p = col_plan(M, cs);
  if (gcs > 2 * C) hipMemsetAsync(dgb, 0, (size_t)M * gcs * sizeof(float), s);   // ragged 2C: keep the padding lanes zero
  ColArgs A = {};
  A.x = x;
  A.dy = dy;
  A.y = y;
  A.a = a;
  A.b = b;
  A.gb = gb;
  A.dgb = dgb;
  A.dxh = dxh;
  A.part = (float*)ws;
  A.M = M;
  A.C = C;
  A.cs = cs;
  A.gcs = gcs;
  A.act = act;
  A.slope = slope;
  launch_col(2, A, p, s);
  col_reduce_kernel<<<cdiv(cs, 16), 256, 0, s>>>((const float*)ws, nullptr, sums, C, cs, p.nb, 0, 0.f);
  return cat::check_launch("spade_bwd_stats");
}

int cat_spade_bwd_apply(const float* x, const float* a, const float* b, const float* sums, double count, float* dx, int64_t M, int C, int cs,
                        cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && M > 0 && count > 0, "spade_bwd_apply: bad geometry");
  cat::ProfScope prof("spade_bwd", 0.0, 12.0 * (double)M * cs, stream);
  const int64_t nquads = M * (cs / 4);
  spade_bwd_apply_kernel<<<ew_grid(nquads), 256, 0, (hipStream_t)stream>>>(x, a, b, sums, (float)(1.0 / count), dx, nquads, cs / 4, cs);
  return cat::check_launch("spade_bwd_apply");
}

int cat_interp_nearest_fwd(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, int C, int xcs, int ycs, cat_stream_t stream) {
  CAT_REQUIRE(xcs % 4 == 0 && ycs % 4 == 0 && ycs >= C && xcs >= ycs && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0,
              "interp_nearest: bad geometry");
  cat::ProfScope prof("interp_nearest", 0.0, 8.0 * (double)N * Ho * Wo * ycs, stream);
  const int64_t total = (int64_t)N * Ho * Wo * (ycs / 4);
  interp_nearest_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, y, total, Hi, Wi, Ho, Wo, ycs / 4, xcs, ycs, (float)Hi / (float)Ho,
                                                                         (float)Wi / (float)Wo);
  return cat::check_launch("interp_nearest");
}

int cat_upsample_nearest_bwd(const float* dy, float* dx, int N, int Hi, int Wi, int f, int C, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && f >= 1 && N > 0, "upsample_nearest_bwd: bad geometry");
  cat::ProfScope prof("upsample_bwd", 0.0, 4.0 * (double)N * Hi * Wi * cs * (f * f + 1), stream);
  const int64_t total = (int64_t)N * Hi * Wi * (cs / 4);
  upsample_bwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(dy, dx, total, Hi, Wi, f, cs / 4, cs);
  return cat::check_launch("upsample_nearest_bwd");
}

int cat_avgpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && N > 0 && H > 0 && W > 0, "avgpool: bad geometry");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  cat::ProfScope prof("avgpool", 0.0, 4.0 * (double)N * cs * ((double)H * W + (double)Ho * Wo), stream);
  const int64_t total = (int64_t)N * Ho * Wo * (cs / 4);
  avgpool_fwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, y, total, H, W, Ho, Wo, cs / 4, cs);
  return cat::check_launch("avgpool_fwd");
}

int cat_avgpool3x3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && N > 0 && H > 0 && W > 0, "avgpool: bad geometry");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  cat::ProfScope prof("avgpool", 0.0, 4.0 * (double)N * cs * ((double)H * W + (double)Ho * Wo), stream);
  const int64_t total = (int64_t)N * H * W * (cs / 4);
  avgpool_bwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(dy, dx, total, H, W, Ho, Wo, cs / 4, cs);
  return cat::check_launch("avgpool_bwd");
}

int cat_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && N > 0 && H >= 2 && W >= 2, "maxpool: bad geometry");
  const int Ho = H / 2, Wo = W / 2;
  cat::ProfScope prof("maxpool", 0.0, 5.0 * (double)N * Ho * Wo * cs * 4.0, stream);
  const int64_t total = (int64_t)N * Ho * Wo * (cs / 4);
  maxpool_fwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, y, total, H, W, Ho, Wo, cs / 4, cs);
  return cat::check_launch("maxpool_fwd");
}

int cat_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && N > 0 && H >= 2 && W >= 2, "maxpool: bad geometry");
  const int Ho = H / 2, Wo = W / 2;
  cat::ProfScope prof("maxpool", 0.0, 9.0 * (double)N * Ho * Wo * cs * 4.0, stream);
  hipStream_t s = (hipStream_t)stream;
  if ((H & 1) || (W & 1)) hipMemsetAsync(dx, 0, (size_t)N * H * W * cs * sizeof(float), s);   // odd edge rows / columns get no gradient
  const int64_t total = (int64_t)N * Ho * Wo * (cs / 4);
  maxpool_bwd_kernel<<<ew_grid(total), 256, 0, s>>>(x, dy, dx, total, H, W, Ho, Wo, cs / 4, cs);
  return cat::check_launch("maxpool_bwd");
}

int cat_onehot_edges(const int* label, const int* inst, float* y, int N, int H, int W, int nc, int cs, cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= nc + (inst ? 1 : 0) && N > 0 && label, "onehot_edges: bad geometry");
  const int64_t total = (int64_t)N * H * W * (cs / 4);
  onehot_edges_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(label, inst, y, total, H, W, nc, cs / 4, cs);
  return cat::check_launch("onehot_edges");
}

size_t cat_spectral_norm_ws_bytes(int O, int I, int taps, int wcs) { return ((size_t)(SN_OSPLIT + 1) * taps * wcs + O + 1024) * sizeof(float); }

int cat_spectral_norm_fwd(const float* w, int O, int I, int taps, int wcs, float* u, float* v, int power_iter, float eps, float* sigma,
                          float* w_sn, float* vp, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(wcs % 4 == 0 && wcs >= I && O > 0 && taps > 0 && u && v && sigma && w_sn && vp && ws, "spectral_norm_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int Kp = taps * wcs;
  float* t = (float*)ws;                    // [SN_OSPLIT][Kp]
  float* sv = t + (size_t)SN_OSPLIT * Kp;   // [O]
  if (power_iter) {
    sn_wt_u_kernel<<<dim3(cdiv(Kp, 256), SN_OSPLIT), 256, 0, s>>>(w, u, t, O, Kp);
    sn_norm_v_kernel<<<1, 1024, 0, s>>>(t, vp, v, I, taps, wcs, eps);
    sn_w_v_kernel<<<O, 256, 0, s>>>(w, vp, sv, Kp);
    sn_norm_u_kernel<<<1, 1024, 0, s>>>(sv, u, sigma, O, eps);
  } else {
    sn_vp_kernel<<<cdiv(Kp, 256), 256, 0, s>>>(v, vp, I, taps, wcs);
    sn_w_v_kernel<<<O, 256, 0, s>>>(w, vp, sv, Kp);
    sn_sigma_kernel<<<1, 1024, 0, s>>>(sv, u, sigma, O);
  }
  const int64_t n4 = (int64_t)O * Kp / 4;
  sn_scale_kernel<<<ew_grid(n4), 256, 0, s>>>(w, sigma, w_sn, n4);
  return cat::check_launch("spectral_norm_fwd");
}

int cat_spectral_norm_bwd(const float* gw, const float* w_sn, const float* u, const float* vp, const float* sigma, int O, int taps, int wcs,
                          float* dw, int accumulate, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(wcs % 4 == 0 && O > 0 && taps > 0 && ws && dw, "spectral_norm_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)O * taps * wcs;
  int npart = (int)((n + 256 * 64 - 1) / (256 * 64));
  if (npart > 1024) npart = 1024;
  sn_dot_kernel<<<npart, 256, 0, s>>>(gw, w_sn, (float*)ws, n);
  sn_bwd_kernel<<<ew_grid(n), 256, 0, s>>>(gw, (const float*)ws, npart, u, vp, sigma, dw, n, taps * wcs, accumulate);
  return cat::check_launch("spectral_norm_bwd");
}

}  // extern "C"
