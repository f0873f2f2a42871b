// InstanceNorm2d / BatchNorm2d (training statistics) fused with the following activation, NHWC, gfx950.
// HBM-bound: forward = 2 reads + 1 write of the activation, backward = 4 reads + 1 write.
//   pass 1  per-(group, channel) shifted sums   sum(x - x0), sum((x - x0)^2)   -> deterministic partials in ws
//   pass 2  finalize: mean / rstd (biased var, eps inside the sqrt), running stats, per-(group,channel) scale+shift
//   pass 3  y = act(x * scale + shift), float4 per lane
// Reduction layout: a workgroup walks pixels with (256 / quads) pixel-lanes x quads channel-quads, so every global
// access is a float4 and a wave touches whole pixels (cs*4 contiguous bytes); cross-lane combine goes through LDS.
#include "common.h"
#include <stdlib.h>

namespace {
using cat::cdiv;

struct NormPlan {
  int G, Pg, nq, nz, zq, ppl, nb;
  size_t part_off, scale_off, shift_off, c1_off, c2_off, bytes;
};

NormPlan plan(const cat_norm_t* g) {
  NormPlan p;
  p.G = g->mode == CAT_NORM_INSTANCE ? g->N : 1;
  p.Pg = g->mode == CAT_NORM_INSTANCE ? g->HW : g->N * g->HW;
  p.nq = g->cs / 4;
  p.nz = cdiv(p.nq, 256);
  p.zq = cdiv(p.nq, p.nz);  // quads per z-block (<= 256)
  p.ppl = 256 / p.zq;
  int nb = cdiv(2048, p.G * p.nz);
  if (nb > 1024) nb = 1024;
  const int maxb = cdiv(p.Pg, p.ppl * 16);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  p.nb = nb;
  size_t off = 0;
  p.part_off = off; off += (size_t)p.G * nb * 2 * g->cs;
  p.scale_off = off; off += (size_t)p.G * g->cs;
  p.shift_off = off; off += (size_t)p.G * g->cs;
  p.c1_off = off; off += (size_t)p.G * g->cs;
  p.c2_off = off; off += (size_t)p.G * g->cs;
  p.bytes = off * sizeof(float);
  return p;
}

// MODE 0: forward stats of x.  MODE 1: backward stats: sum(g), sum(g * xhat) with g = dy * act'(pre).
template <int MODE>
__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ part, int Pg, int C, int cs, int zq, int ppl, int nb,
                                                         int act, float slope) {
  __shared__ f4 red[2][256];
  const int g = blockIdx.y, b = blockIdx.x, z = blockIdx.z;
  const int tid = threadIdx.x;
  const int cq_l = tid % zq, pl = tid / zq;
  const int cq = z * zq + cq_l;
  const bool active = pl < ppl && cq * 4 < cs;
  const int per = (Pg + nb - 1) / nb;
  const int pbeg = b * per, pend = min(Pg, pbeg + per);
  const float* xg = x + (int64_t)g * Pg * cs;
  f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int c = cq * 4;
    if (MODE == 0) {
      const f4 sh = *reinterpret_cast<const f4*>(xg + c);
      for (int p0 = pbeg + pl; p0 < pend; p0 += 4 * ppl) {
        f4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = p0 + k * ppl;
          q[k] = p < pend ? *reinterpret_cast<const f4*>(xg + (int64_t)p * cs + c) : sh;      // x - x0 = 0: adds nothing
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f4 v = q[k] - sh;
          s0 += v;
          s1 += v * v;
        }
      }
    } else {
      const float* dg = dy + (int64_t)g * Pg * cs;
      f4 mu, rs, ga, be;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cv = c + e < C;
        mu[e] = cv ? mean[g * C + c + e] : 0.f;
        rs[e] = cv ? rstd[g * C + c + e] : 0.f;
        ga[e] = cv ? (gamma ? gamma[c + e] : 1.f) : 0.f;
        be[e] = cv ? (beta ? beta[c + e] : 0.f) : 0.f;
      }
      for (int p0 = pbeg + pl; p0 < pend; p0 += 4 * ppl) {      // four independent pixels per iteration: more loads in flight
        f4 xq[4], gq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = p0 + k * ppl;
          const bool v = p < pend;
          xq[k] = v ? *reinterpret_cast<const f4*>(xg + (int64_t)p * cs + c) : mu;     // xhat = 0, g = 0: adds nothing
          gq[k] = v ? *reinterpret_cast<const f4*>(dg + (int64_t)p * cs + c) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // same summation order as the one-pixel loop
          const f4 xh = (xq[k] - mu) * rs;
          f4 gv = gq[k];
          if (act != CAT_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] *= cat::act_grad_from_out(cat::apply_act(ga[e] * xh[e] + be[e], act, slope), act, slope);
          }
          s0 += gv;
          s1 += gv * xh;
        }
      }
    }
  }
  red[0][tid] = s0;
  red[1][tid] = s1;
  __syncthreads();
  if (pl == 0 && cq * 4 < cs) {
    for (int j = 1; j < ppl; ++j) {
      s0 += red[0][j * zq + cq_l];
      s1 += red[1][j * zq + cq_l];
    }
    float* dst = part + ((int64_t)(g * nb + b) * 2) * cs + cq * 4;
    *reinterpret_cast<f4*>(dst) = s0;
    *reinterpret_cast<f4*>(dst + cs) = s1;
  }
}

__global__ __launch_bounds__(256) void norm_fwd_finalize_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float* __restrict__ scale, float* __restrict__ shift, int G, int Pg,
                                                                int C, int cs, int nb, float eps, float momentum,
                                                                int64_t* __restrict__ num_batches) {
  // grid (cdiv(cs,4), G); one WAVE per channel: 64 lanes stride over the nb partial blocks, wave-shuffle reduce (the serial
  // 64-channels x 4-lanes version cost 55 us at nb = 1024 -- more than the stats pass it finishes)
  const int g = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (num_batches && blockIdx.x == 0 && g == 0 && threadIdx.x == 0) *num_batches += 1;   // nn.BatchNorm2d.num_batches_tracked
  if (c >= cs) return;
  const int idx = g * cs + c;
  if (c >= C) {
    if (lane == 0) {
      scale[idx] = 0.f;
      shift[idx] = 0.f;
    }
    return;
  }
  float s0 = 0.f, s1 = 0.f;
  for (int b = lane; b < nb; b += 64) {
    const float* src = part + ((int64_t)(g * nb + b) * 2) * cs + c;
    s0 += src[0];
    s1 += src[cs];
  }
  s0 = cat::wave_sum(s0);
  s1 = cat::wave_sum(s1);
  if (lane != 0) return;
  const float inv = 1.f / (float)Pg;
  const float d = s0 * inv;
  const float mean = x[(int64_t)g * Pg * cs + c] + d;
  float var = s1 * inv - d * d;
  var = var > 0.f ? var : 0.f;
  const float rstd = rsqrtf(var + eps);
  save_mean[g * C + c] = mean;
  save_rstd[g * C + c] = rstd;
  if (running_mean) {  // batch norm only (G == 1)
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    const float unb = Pg > 1 ? var * (float)Pg / (float)(Pg - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  scale[idx] = ga * rstd;
  shift[idx] = be - mean * ga * rstd;
}

// y = act(x * scale[g][c] + shift[g][c]); scale/shift hold zeros on padding channels.
// IT = index type of the element walk: int64_t, or int (kept switched off: `idx32`'s `on`) when the tensor has < 2^31 quads -- the `%` and `/` below are
// emulated in ~100 instructions each at 64 bits, which is most of what these HBM-bound kernels execute per float4.
template <typename IT>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, IT nquads, int nq,
                                                         IT qpg, int cs, int act, float slope) {
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (IT)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    const int g = (int)(i / qpg);
    const f4 v = *reinterpret_cast<const f4*>(x + (int64_t)i * 4);
    const f4 sc = *reinterpret_cast<const f4*>(scale + g * cs + cq * 4);
    const f4 sh = *reinterpret_cast<const f4*>(shift + g * cs + cq * 4);
    f4 o = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = cat::apply_act(o[e], act, slope);
    *reinterpret_cast<f4*>(y + (int64_t)i * 4) = o;
  }
}

__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                const float* __restrict__ rstd, float* __restrict__ c1,
                                                                float* __restrict__ c2, float* __restrict__ scale, int Pg, int C, int cs,
                                                                int nb, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int accumulate) {
  // grid (cdiv(cs,4), G); one wave per channel (see norm_fwd_finalize_kernel): per-(group, channel) means of g and g*xhat
  const int g = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= cs) return;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    for (int b = lane; b < nb; b += 64) {
      const float* src = part + ((int64_t)(g * nb + b) * 2) * cs + c;
      s0 += src[0];
      s1 += src[cs];
    }
  }
  s0 = cat::wave_sum(s0);
  s1 = cat::wave_sum(s1);
  if (lane != 0) return;
  const float inv = 1.f / (float)Pg;
  c1[g * cs + c] = s0 * inv;
  c2[g * cs + c] = s1 * inv;
  scale[g * cs + c] = c < C ? (gamma ? gamma[c] : 1.f) * rstd[g * C + c] : 0.f;
  if (c < C) {   // one statistic group (batch norm): the parameter gradients are these sums -- no separate launch
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s1 : s1;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s0 : s0;
  }
}

// nn.InstanceNorm2d(track_running_stats=True).train() (models/networks.py:29-64 with --norm instance --norm_track_running_stats): torch
// feeds the instances to batch_norm as N * C channels and averages the per-instance updates over the batch, i.e.
//   running_mean[c] = (1 - m) * running_mean[c] + m * mean_n(mean[n][c]),  running_var[c] likewise with the UNBIASED per-instance variance
// (recovered from the saved rstd: var = 1 / rstd^2 - eps).
__global__ __launch_bounds__(256) void in_running_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var, int G, int Pg,
                                                         int C, float eps, float momentum, int64_t* __restrict__ num_batches) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (num_batches && c == 0) *num_batches += 1;
  if (c >= C) return;
  float sm = 0.f, sv = 0.f;
  for (int g = 0; g < G; ++g) {
    const float r = rstd[g * C + c];
    float var = 1.f / (r * r) - eps;
    var = var > 0.f ? var : 0.f;
    sm += mean[g * C + c];
    sv += Pg > 1 ? var * (float)Pg / (float)(Pg - 1) : var;
  }
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (sm / (float)G);
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (sv / (float)G);
}

// dgamma[c] (+)= sum_g sum(g*xhat), dbeta[c] (+)= sum_g sum(g)  (sums recovered from the per-group means)
__global__ __launch_bounds__(256) void norm_bwd_param_kernel(const float* __restrict__ c1, const float* __restrict__ c2,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int G, int Pg, int C,
                                                             int cs, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float tb = 0.f, tg = 0.f;
  for (int g = 0; g < G; ++g) {
    tb += c1[g * cs + c];
    tg += c2[g * cs + c];
  }
  tb *= (float)Pg;
  tg *= (float)Pg;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + tg : tg;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + tb : tb;
}

// dx = gamma*rstd * (g - mean(g) - xhat * mean(g*xhat))
template <typename IT>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ c1, const float* __restrict__ c2,
                                                             const float* __restrict__ scale, float* __restrict__ dx, IT nquads,
                                                             int nq, IT qpg, int C, int cs, int act, float slope) {
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (IT)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    const int g = (int)(i / qpg);
    const int c = cq * 4;
    const f4 xv = *reinterpret_cast<const f4*>(x + (int64_t)i * 4);
    f4 gv = *reinterpret_cast<const f4*>(dy + (int64_t)i * 4);
    const f4 m1 = *reinterpret_cast<const f4*>(c1 + g * cs + c);
    const f4 m2 = *reinterpret_cast<const f4*>(c2 + g * cs + c);
    const f4 sc = *reinterpret_cast<const f4*>(scale + g * cs + c);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool cv = c + e < C;
      const float mu = cv ? mean[g * C + c + e] : 0.f, rs = cv ? rstd[g * C + c + e] : 0.f;
      const float xh = (xv[e] - mu) * rs;
      float gg = gv[e];
      if (act != CAT_ACT_NONE) {
        const float ga = cv ? (gamma ? gamma[c + e] : 1.f) : 0.f, be = cv ? (beta ? beta[c + e] : 0.f) : 0.f;
        gg *= cat::act_grad_from_out(cat::apply_act(ga * xh + be, act, slope), act, slope);
      }
      o[e] = sc[e] * (gg - m1[e] - xh * m2[e]);
    }
    *reinterpret_cast<f4*>(dx + (int64_t)i * 4) = o;
  }
}

// Pixel-walk variants of the two apply passes: a thread owns ONE channel quad (its per-(group, channel) coefficients live in registers for the
// whole walk) and strides over the pixels of its block, four independent pixels per iteration -- no per-element index division and no
// per-element coefficient loads (the element-walk kernels issue 5 - 13 loads per float4 of payload), more loads in flight per wave.
__global__ __launch_bounds__(256) void norm_apply_walk_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y, int Pg, int cs, int zq,
                                                              int ppl, int nb, int act, float slope) {
  const int g = blockIdx.y, b = blockIdx.x, z = blockIdx.z;
  const int cq = z * zq + threadIdx.x % zq, pl = threadIdx.x / zq;
  if (pl >= ppl || cq * 4 >= cs) return;
  const int per = (Pg + nb - 1) / nb;
  const int pbeg = b * per, pend = min(Pg, pbeg + per);
  const int c = cq * 4;
  const f4 sc = *reinterpret_cast<const f4*>(scale + g * cs + c), sh = *reinterpret_cast<const f4*>(shift + g * cs + c);
  const float* xg = x + (int64_t)g * Pg * cs + c;
  float* yg = y + (int64_t)g * Pg * cs + c;
  for (int p0 = pbeg + pl; p0 < pend; p0 += 4 * ppl) {
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + k * ppl;
      v[k] = p < pend ? *reinterpret_cast<const f4*>(xg + (int64_t)p * cs) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + k * ppl;
      f4 o = v[k] * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = cat::apply_act(o[e], act, slope);
      if (p < pend) *reinterpret_cast<f4*>(yg + (int64_t)p * cs) = o;
    }
  }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_walk_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ c1, const float* __restrict__ c2,
                                                                  const float* __restrict__ scale, float* __restrict__ dx, int Pg, int C, int cs,
                                                                  int zq, int ppl, int nb, int act, float slope) {
  const int g = blockIdx.y, b = blockIdx.x, z = blockIdx.z;
  const int cq = z * zq + threadIdx.x % zq, pl = threadIdx.x / zq;
  if (pl >= ppl || cq * 4 >= cs) return;
  const int per = (Pg + nb - 1) / nb;
  const int pbeg = b * per, pend = min(Pg, pbeg + per);
  const int c = cq * 4;
  const f4 m1 = *reinterpret_cast<const f4*>(c1 + g * cs + c), m2 = *reinterpret_cast<const f4*>(c2 + g * cs + c);
  const f4 sc = *reinterpret_cast<const f4*>(scale + g * cs + c);
  f4 mu, rs, ga, be;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool cv = c + e < C;
    mu[e] = cv ? mean[g * C + c + e] : 0.f;
    rs[e] = cv ? rstd[g * C + c + e] : 0.f;
    ga[e] = cv ? (gamma ? gamma[c + e] : 1.f) : 0.f;
    be[e] = cv ? (beta ? beta[c + e] : 0.f) : 0.f;
  }
  const float* xg = x + (int64_t)g * Pg * cs + c;
  const float* dg = dy + (int64_t)g * Pg * cs + c;
  float* og = dx + (int64_t)g * Pg * cs + c;
  for (int p0 = pbeg + pl; p0 < pend; p0 += 4 * ppl) {
    f4 xv[4], gv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + k * ppl;
      const bool v = p < pend;
      xv[k] = v ? *reinterpret_cast<const f4*>(xg + (int64_t)p * cs) : f4{0.f, 0.f, 0.f, 0.f};
      gv[k] = v ? *reinterpret_cast<const f4*>(dg + (int64_t)p * cs) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + k * ppl;
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[k][e] - mu[e]) * rs[e];
        float gg = gv[k][e];
        if (act != CAT_ACT_NONE) gg *= cat::act_grad_from_out(cat::apply_act(ga[e] * xh + be[e], act, slope), act, slope);
        o[e] = sc[e] * (gg - m1[e] - xh * m2[e]);
      }
      if (p < pend) *reinterpret_cast<f4*>(og + (int64_t)p * cs) = o;
    }
  }
}

// block count of the apply walks: ~4096 workgroups, at least 8 pixels per pixel lane
static int walk_blocks(const NormPlan& p) {
  int nb = cdiv(4096, p.G * p.nz);
  const int maxb = cdiv(p.Pg, p.ppl * 8);
  if (nb > maxb) nb = maxb;
  return nb < 1 ? 1 : nb;
}
static bool walk_on() {
  static const int on = getenv("CAT_NORM_WALK") ? atoi(getenv("CAT_NORM_WALK")) : 1;
  return on != 0;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                               float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = (gamma ? gamma[c] : 1.f) * rsqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * s;
}

template <typename IT>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, IT nquads, int nq,
                                                         int C, int act, float slope) {
  for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (IT)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const f4 v = *reinterpret_cast<const f4*>(x + (int64_t)i * 4);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = c + e < C ? cat::apply_act(v[e] * scale[c + e] + shift[c + e], act, slope) : 0.f;
    *reinterpret_cast<f4*>(y + (int64_t)i * 4) = o;
  }
}

int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// 32-bit element walk (opt-in): the loop variable may run one grid stride (<= 8192 * 256) past nquads
bool idx32(int64_t nquads) {
  constexpr int on = 0;
  return on && nquads < (int64_t)2147483647 - 8192 * 256;
}

}  // namespace

extern "C" {

size_t cat_norm_ws_bytes(const cat_norm_t* g) { return plan(g).bytes; }

int cat_norm_fwd(const cat_norm_t* g, const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                 float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws,
                 cat_stream_t stream) {
  CAT_REQUIRE(g->cs % 4 == 0 && g->cs >= g->C && g->N > 0 && g->HW > 0, "norm: bad geometry");
  CAT_REQUIRE(ws && save_mean && save_rstd, "norm fwd: workspace / save buffers required");
  cat::ProfScope prof("norm_fwd", 0.0, 3 * 4.0 * (double)g->N * g->HW * g->cs, stream);
  const NormPlan p = plan(g);
  float* w = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  norm_stats_kernel<0><<<dim3(p.nb, p.G, p.nz), 256, 0, s>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, w + p.part_off, p.Pg,
                                                              g->C, g->cs, p.zq, p.ppl, p.nb, 0, 0.f);
  norm_fwd_finalize_kernel<<<dim3(cdiv(g->cs, 4), p.G), 256, 0, s>>>(x, w + p.part_off, gamma, beta, save_mean, save_rstd,
                                                                   g->mode == CAT_NORM_BATCH ? running_mean : nullptr,
                                                                   g->mode == CAT_NORM_BATCH ? running_var : nullptr, w + p.scale_off,
                                                                   w + p.shift_off, p.G, p.Pg, g->C, g->cs, p.nb, g->eps, g->momentum,
                                                                   g->mode == CAT_NORM_BATCH ? num_batches_tracked : nullptr);
  if (g->mode == CAT_NORM_INSTANCE && running_mean && running_var)
    in_running_kernel<<<cdiv(g->C, 256), 256, 0, s>>>(save_mean, save_rstd, running_mean, running_var, p.G, p.Pg, g->C, g->eps, g->momentum,
                                                       num_batches_tracked);
  const int64_t nquads = (int64_t)g->N * g->HW * p.nq;
  if (walk_on()) {
    const int nbw = walk_blocks(p);
    norm_apply_walk_kernel<<<dim3(nbw, p.G, p.nz), 256, 0, s>>>(x, w + p.scale_off, w + p.shift_off, y, p.Pg, g->cs, p.zq, p.ppl, nbw, g->act, g->slope);
    return cat::check_launch("norm_fwd");
  }
  if (idx32(nquads))
    norm_apply_kernel<int><<<ew_grid(nquads), 256, 0, s>>>(x, w + p.scale_off, w + p.shift_off, y, (int)nquads, p.nq, p.Pg * p.nq, g->cs,
                                                            g->act, g->slope);
  else
    norm_apply_kernel<int64_t><<<ew_grid(nquads), 256, 0, s>>>(x, w + p.scale_off, w + p.shift_off, y, nquads, p.nq, (int64_t)p.Pg * p.nq,
                                                                g->cs, g->act, g->slope);
  return cat::check_launch("norm_fwd");
}

int cat_norm_bwd(const cat_norm_t* g, const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                 const float* save_rstd, float* dx, float* dgamma, float* dbeta, int accumulate, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(g->cs % 4 == 0 && g->cs >= g->C && g->N > 0 && g->HW > 0, "norm: bad geometry");
  CAT_REQUIRE(ws, "norm bwd: workspace required");
  cat::ProfScope prof("norm_bwd", 0.0, 5 * 4.0 * (double)g->N * g->HW * g->cs, stream);
  const NormPlan p = plan(g);
  float* w = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  norm_stats_kernel<1><<<dim3(p.nb, p.G, p.nz), 256, 0, s>>>(x, dy, gamma, beta, save_mean, save_rstd, w + p.part_off, p.Pg, g->C, g->cs,
                                                              p.zq, p.ppl, p.nb, g->act, g->slope);
  const bool one_group = p.G == 1;
  norm_bwd_finalize_kernel<<<dim3(cdiv(g->cs, 4), p.G), 256, 0, s>>>(w + p.part_off, gamma, save_rstd, w + p.c1_off, w + p.c2_off,
                                                                        w + p.scale_off, p.Pg, g->C, g->cs, p.nb, one_group ? dgamma : nullptr,
                                                                        one_group ? dbeta : nullptr, accumulate);
  if (!one_group && (dgamma || dbeta))
    norm_bwd_param_kernel<<<cdiv(g->C, 256), 256, 0, s>>>(w + p.c1_off, w + p.c2_off, dgamma, dbeta, p.G, p.Pg, g->C, g->cs, accumulate);
  const int64_t nquads = (int64_t)g->N * g->HW * p.nq;
  if (walk_on()) {
    const int nbw = walk_blocks(p);
    norm_bwd_apply_walk_kernel<<<dim3(nbw, p.G, p.nz), 256, 0, s>>>(x, dy, gamma, beta, save_mean, save_rstd, w + p.c1_off, w + p.c2_off,
                                                                     w + p.scale_off, dx, p.Pg, g->C, g->cs, p.zq, p.ppl, nbw, g->act, g->slope);
    return cat::check_launch("norm_bwd");
  }
  if (idx32(nquads))
    norm_bwd_apply_kernel<int><<<ew_grid(nquads), 256, 0, s>>>(x, dy, gamma, beta, save_mean, save_rstd, w + p.c1_off, w + p.c2_off,
                                                                w + p.scale_off, dx, (int)nquads, p.nq, p.Pg * p.nq, g->C, g->cs, g->act,
                                                                g->slope);
  else
    norm_bwd_apply_kernel<int64_t><<<ew_grid(nquads), 256, 0, s>>>(x, dy, gamma, beta, save_mean, save_rstd, w + p.c1_off, w + p.c2_off,
                                                                    w + p.scale_off, dx, nquads, p.nq, (int64_t)p.Pg * p.nq, g->C, g->cs,
                                                                    g->act, g->slope);
  return cat::check_launch("norm_bwd");
}

int cat_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                float* scale, float* shift, cat_stream_t stream) {
  bn_fold_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, C, scale, shift);
  return cat::check_launch("bn_fold");
}

int cat_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int64_t M, int C, int cs, int act, float slope,
                       cat_stream_t stream) {
  CAT_REQUIRE(cs % 4 == 0 && cs >= C, "affine_act: bad channel stride");
  const int64_t nquads = M * (cs / 4);
  cat::ProfScope prof("affine_act", 0.0, 8.0 * M * cs, stream);
  if (idx32(nquads)) affine_act_kernel<int><<<ew_grid(nquads), 256, 0, (hipStream_t)stream>>>(x, scale, shift, y, (int)nquads, cs / 4, C, act, slope);
  else affine_act_kernel<int64_t><<<ew_grid(nquads), 256, 0, (hipStream_t)stream>>>(x, scale, shift, y, nquads, cs / 4, C, act, slope);
  return cat::check_launch("affine_act");
}

}  // extern "C"
