// Inference-only layers of the evaluation path: the FID feature extractor InceptionV3 (metric/inception.py:16-150,
// 177-300 over torchvision 0.8.2's Inception3) needs, besides convolutions, three pooling flavours and the bilinear input resize
// (F.interpolate(x, (299, 299), mode='bilinear', align_corners=False), metric/inception.py:129-133).  NHWC fp32, gfx950.  All HBM-bound
// element walks: one thread = one (output pixel, channel quad) float4; channel-slice outputs (ycs > C4, y pointing at the slice) let the four
// branches of an Inception block write straight into the concatenated tensor (torch.cat(outputs, 1) never runs).
#include "common.h"

namespace {

constexpr int POOL_MAX = 0, POOL_AVG_EXCL = 1;

// y[n][oy][ox][c] = max / mean over the window rows iy = oy * stride - pad + ky, taps outside the plane excluded
// (nn.MaxPool2d: -inf padding; F.avg_pool2d(count_include_pad=False): divisor = taps inside)
__global__ __launch_bounds__(256) void pool2d_kernel(const float* __restrict__ x, int xcs, int N, int H, int W, int nq, int k, int stride,
                                                     int pad, int mode, float* __restrict__ y, int ycs, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t r = i / nq;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    f4 acc = mode == POOL_MAX ? f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f4{0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = iy0 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ix0 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f4 v = *reinterpret_cast<const f4*>(x + (((int64_t)n * H + iy) * W + ix) * xcs + q * 4);
        if (mode == POOL_MAX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], v[e]);
        } else {
          acc += v;
        }
        ++cnt;
      }
    }
    if (mode == POOL_AVG_EXCL) acc = acc / (float)(cnt > 0 ? cnt : 1);
    *reinterpret_cast<f4*>(y + (((int64_t)n * Ho + oy) * Wo + ox) * ycs + q * 4) = acc;
  }
}

// y[n][c] = mean over the plane (nn.AdaptiveAvgPool2d((1, 1))): one wave per (image, channel quad), pixels dealt to the lanes, fixed
// shuffle tree -> deterministic
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, int xcs, int N, int HW, int nq, float* __restrict__ y,
                                                             int ycs) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= N * nq) return;
  const int n = item / nq, q = item - n * nq;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = lane; p < HW; p += 64) acc += *reinterpret_cast<const f4*>(x + ((int64_t)n * HW + p) * xcs + q * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = cat::wave_sum(acc[e]);
  if (lane == 0) *reinterpret_cast<f4*>(y + (int64_t)n * ycs + q * 4) = acc / (float)HW;
}

// torch's upsample_bilinear2d, align_corners = False, no antialias: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out;
// neighbours i0 = floor(src), i1 = min(i0 + 1, in - 1), weight l1 = src - i0.  Output = a * value + b (the `2 * x - 1` input
// normalisation of metric/inception.py:135-136 fused into the same pass).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int xcs, int N, int H, int W, int C, int nq,
                                                              float* __restrict__ y, int ycs, int Ho, int Wo, float sh, float sw, float a,
                                                              float b) {
  const int64_t total = (int64_t)N * Ho * Wo * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % nq);
    int64_t r = i / nq;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float sy = sh * ((float)oy + 0.5f) - 0.5f, sx = sw * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* base = x + (int64_t)n * H * W * xcs + q * 4;
    const f4 v00 = *reinterpret_cast<const f4*>(base + ((int64_t)y0 * W + x0) * xcs), v01 = *reinterpret_cast<const f4*>(base + ((int64_t)y0 * W + x1) * xcs);
    const f4 v10 = *reinterpret_cast<const f4*>(base + ((int64_t)y1 * W + x0) * xcs), v11 = *reinterpret_cast<const f4*>(base + ((int64_t)y1 * W + x1) * xcs);
    f4 v = a * (ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11)) + b;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = q * 4 + e < C ? v[e] : 0.f;      // padding channels stay exactly 0
    *reinterpret_cast<f4*>(y + (((int64_t)n * Ho + oy) * Wo + ox) * ycs + q * 4) = v;
  }
}

int walk_grid(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" {

int cat_pool2d_fwd(const float* x, int xcs, int N, int H, int W, int C4, int k, int stride, int pad, int mode, float* y, int ycs, int Ho,
                   int Wo, cat_stream_t stream) {
  CAT_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C4 > 0 && (C4 & 3) == 0 && (xcs & 3) == 0 && (ycs & 3) == 0 && xcs >= C4 && ycs >= C4,
              "pool2d: channel layout");
  CAT_REQUIRE(k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && 2 * pad <= k && (mode == POOL_MAX || mode == POOL_AVG_EXCL), "pool2d: window");
  CAT_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1 && Ho > 0 && Wo > 0, "pool2d: output size");
  cat::ProfScope prof("pool2d", 0.0, 4.0 * ((double)N * H * W + (double)N * Ho * Wo) * C4, stream);
  pool2d_kernel<<<walk_grid((int64_t)N * Ho * Wo * (C4 / 4)), 256, 0, (hipStream_t)stream>>>(x, xcs, N, H, W, C4 / 4, k, stride, pad, mode, y, ycs, Ho, Wo);
  return cat::check_launch("pool2d");
}

int cat_global_avgpool_fwd(const float* x, int xcs, int N, int HW, int C4, float* y, int ycs, cat_stream_t stream) {
  CAT_REQUIRE(x && y && N > 0 && HW > 0 && C4 > 0 && (C4 & 3) == 0 && (xcs & 3) == 0 && (ycs & 3) == 0 && xcs >= C4 && ycs >= C4,
              "global avgpool: channel layout");
  cat::ProfScope prof("global_avgpool", 0.0, 4.0 * (double)N * HW * C4, stream);
  global_avgpool_kernel<<<cat::cdiv((int64_t)N * (C4 / 4), 4), 256, 0, (hipStream_t)stream>>>(x, xcs, N, HW, C4 / 4, y, ycs);
  return cat::check_launch("global_avgpool");
}

int cat_resize_bilinear_fwd(const float* x, int xcs, int N, int H, int W, int C, float* y, int ycs, int Ho, int Wo, float a, float b,
                            cat_stream_t stream) {
  const int C4 = (C + 3) & ~3;
  CAT_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && (xcs & 3) == 0 && (ycs & 3) == 0 && xcs >= C4 && ycs >= C4,
              "resize bilinear: layout");
  cat::ProfScope prof("resize_bilinear", 0.0, 4.0 * ((double)N * H * W + (double)N * Ho * Wo) * C4, stream);
  resize_bilinear_kernel<<<walk_grid((int64_t)N * Ho * Wo * (C4 / 4)), 256, 0, (hipStream_t)stream>>>(x, xcs, N, H, W, C, C4 / 4, y, ycs, Ho, Wo, (float)H / (float)Ho,
                                                                                                      (float)W / (float)Wo, a, b);
  return cat::check_launch("resize_bilinear");
}

}  // extern "C"
