// HBM-bound elementwise / layout / optimizer kernels (gfx950).  All of them move float4 per lane over flat NHWC
// buffers (pixel stride is a multiple of 4 floats) and grid-stride over at most 8192 workgroups.
#include "common.h"
#include <stdint.h>

namespace {
using cat::cdiv;

int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t nq, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
    f4 v = *reinterpret_cast<const f4*>(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = cat::apply_act(v[e], act, slope);
    *reinterpret_cast<f4*>(y + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                                      int64_t nq, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
    const f4 o = *reinterpret_cast<const f4*>(y + i * 4);
    f4 g = *reinterpret_cast<const f4*>(dy + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= cat::act_grad_from_out(o[e], act, slope);
    *reinterpret_cast<f4*>(dx + i * 4) = g;
  }
}

struct AddSrcs {
  const float* p[8];
};
template <int NS>
__global__ __launch_bounds__(256) void add_n_kernel(AddSrcs s, float* __restrict__ dst, int64_t nq) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
    f4 v = *reinterpret_cast<const f4*>(s.p[0] + i * 4);
#pragma unroll
    for (int k = 1; k < NS; ++k) v += *reinterpret_cast<const f4*>(s.p[k] + i * 4);
    *reinterpret_cast<f4*>(dst + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int ca, int acs, const float* __restrict__ b, int cb,
                                                      int bcs, float* __restrict__ y, int ycs, int64_t M) {
  const int64_t total = M * ycs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / ycs;
    const int c = (int)(i - m * ycs);
    float v = 0.f;
    if (c < ca) v = a[m * acs + c];
    else if (c < ca + cb) v = b[m * bcs + (c - ca)];
    y[i] = v;
  }
}

__global__ __launch_bounds__(256) void slice_kernel(const float* __restrict__ x, int xcs, int c0, int c, float* __restrict__ y, int ycs,
                                                    int64_t M) {
  const int64_t total = M * ycs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / ycs;
    const int cc = (int)(i - m * ycs);
    y[i] = cc < c ? x[m * xcs + c0 + cc] : 0.f;
  }
}

// NCHW -> NHWC(cs): tile transpose through LDS: 64 pixels x 64 channels per workgroup.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int ycs) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? x[((int64_t)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int p = p0 + j, c = c0 + tx;
    if (p < HW && c < ycs) y[((int64_t)n * HW + p) * ycs + c] = tile[tx][j];
  }
}

// C <= 4 into a 4-wide pixel stride (the RGB batches of set_input): one lane per pixel, C coalesced plane reads, one float4 store -- the
// 64 x 64 tile transpose above spends 61 of 64 rows / lanes on padding there (36 us for a 16 x 3 x 256 x 256 batch; this form: HBM time)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / HW;
    const int p = (int)(i - n * HW);
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < C) v[e] = x[(n * C + e) * HW + p];
    *reinterpret_cast<f4*>(y + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int xcs) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && p < HW) ? x[((int64_t)n * HW + p) * xcs + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) y[((int64_t)n * C + c) * HW + p] = tile[tx][j];
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] += a * x[i];
}

// torch.optim.Adam (no amsgrad): identical operation order to torch/optim/adam.py::_single_tensor_adam.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float b1, float b2, float eps, float wd,
                                                   float step_size, float inv_bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);  // lerp, as torch does
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// Device-resident optimiser state (hipGraph-replayable step): h = [lr, beta1, beta2, eps, weight_decay, step, step_size, 1/sqrt(bc2)].
// One thread advances the step counter and derives the bias corrections exactly as cat_adam_step does on the host.
__global__ void adam_tick_kernel(float* __restrict__ h) {
  const double step = (double)h[5] + 1.0;
  const double bc1 = 1.0 - pow((double)h[1], step);
  const double bc2 = 1.0 - pow((double)h[2], step);
  h[5] = (float)step;
  h[6] = (float)((double)h[0] / bc1);
  h[7] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const float* __restrict__ h, float gscale) {
  const float b1 = h[1], b2 = h[2], eps = h[3], wd = h[4], step_size = h[6], inv_bc2_sqrt = h[7];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// per-channel sums over pixels, two deterministic stages
__global__ __launch_bounds__(256) void chansum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t M, int cs,
                                                              int zq, int ppl, int nb) {
  __shared__ f4 red[256];
  const int b = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
  const int cq_l = tid % zq, pl = tid / zq, cq = z * zq + cq_l;
  const int64_t per = (M + nb - 1) / nb;
  const int64_t pbeg = b * per, pend = pbeg + per < M ? pbeg + per : M;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (pl < ppl && cq * 4 < cs)
    for (int64_t p = pbeg + pl; p < pend; p += ppl) s += *reinterpret_cast<const f4*>(x + p * cs + cq * 4);
  red[tid] = s;
  __syncthreads();
  if (pl == 0 && cq * 4 < cs) {
    for (int j = 1; j < ppl; ++j) s += red[j * zq + cq_l];
    *reinterpret_cast<f4*>(part + (int64_t)b * cs + cq * 4) = s;
  }
}
// 16 channels x 16 partial lanes per workgroup, four independent loads per lane and step, lanes combined pairwise through LDS in a fixed order
// (round 6: 64 channels x 4 lanes walked up to 128 dependent loads per lane: 16 us for 64 KB of partials)
__global__ __launch_bounds__(256) void chansum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int cs, int nb,
                                                            int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), zl = threadIdx.x >> 4;
  float s = 0.f;
  if (c < C) {
    const float* src = part + c;
    int b = zl;
    for (; b + 48 < nb; b += 64) {
      const float v0 = src[(int64_t)b * cs], v1 = src[(int64_t)(b + 16) * cs], v2 = src[(int64_t)(b + 32) * cs], v3 = src[(int64_t)(b + 48) * cs];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; b < nb; b += 16) s += src[(int64_t)b * cs];
  }
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int st = 8; st >= 1; st >>= 1) {
    if (zl < st) red[threadIdx.x] += red[threadIdx.x + st * 16];
    __syncthreads();
  }
  if (zl == 0 && c < C) out[c] = accumulate ? out[c] + red[threadIdx.x] : red[threadIdx.x];
}

struct CsPlan { int nq, nz, zq, ppl, nb; };
CsPlan cs_plan(int64_t M, int cs) {
  CsPlan p;
  p.nq = cs / 4;
  p.nz = cdiv(p.nq, 256);
  p.zq = cdiv(p.nq, p.nz);
  p.ppl = 256 / p.zq;
  int nb = cdiv(512, p.nz);
  const int maxb = cdiv(M, (int64_t)p.ppl * 16);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  p.nb = nb;
  return p;
}

// backward of ReflectionPad2d: every source pixel gathers the (up to 3 x 3) padded positions that mirror onto it
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int N, int H, int W,
                                                           int cs, int pad) {
  const int nq = cs / 4;
  const int64_t total = (int64_t)N * H * W * nq;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
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
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += *reinterpret_cast<const f4*>(dxp + (((int64_t)n * Hp + ys[a]) * Wp + xs[b]) * cs + cq * 4);
    *reinterpret_cast<f4*>(dx + i * 4) = s;
  }
}

// nn.ReplicationPad2d(pad) (reference inception_modules.py:114-115, padding_type='replicate'): the padded copy is materialised (the option is
// not used by any launch script; the convolution behind it then runs with padding 0) and its backward folds the clamped positions
__global__ __launch_bounds__(256) void replicate_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int cs,
                                                                int pad) {
  const int nq = cs / 4;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t total = (int64_t)N * Hp * Wp * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
    const int xp = (int)(r % Wp);
    r /= Wp;
    const int yp = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int ys = min(max(yp - pad, 0), H - 1), xs = min(max(xp - pad, 0), W - 1);
    *reinterpret_cast<f4*>(y + i * 4) = *reinterpret_cast<const f4*>(x + (((int64_t)n * H + ys) * W + xs) * cs + cq * 4);
  }
}

// dx[n, y, x] = sum of dyp over the padded positions that clamp onto (y, x): one row / column in the interior, pad + 1 of them on a border
__global__ __launch_bounds__(256) void replicate_pad_bwd_kernel(const float* __restrict__ dyp, float* __restrict__ dx, int N, int H, int W, int cs,
                                                                int pad) {
  const int nq = cs / 4;
  const int Wp = W + 2 * pad, Hp = H + 2 * pad;
  const int64_t total = (int64_t)N * H * W * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % nq);
    int64_t r = i / nq;
    const int xw = (int)(r % W);
    r /= W;
    const int yh = (int)(r % H);
    const int n = (int)(r / H);
    const int y0 = yh == 0 ? 0 : yh + pad, y1 = yh == H - 1 ? Hp - 1 : yh + pad;     // H == 1: the single row collects all of them
    const int x0 = xw == 0 ? 0 : xw + pad, x1 = xw == W - 1 ? Wp - 1 : xw + pad;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = y0; a <= y1; ++a)
      for (int b = x0; b <= x1; ++b) s += *reinterpret_cast<const f4*>(dyp + (((int64_t)n * Hp + a) * Wp + b) * cs + cq * 4);
    *reinterpret_cast<f4*>(dx + i * 4) = s;
  }
}

}  // namespace

extern "C" {

int cat_replicate_pad_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, int pad, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 8.0 * N * (H + 2 * pad) * (W + 2 * pad) * cs, stream);
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && pad >= 0 && N > 0 && H > 0 && W > 0, "replicate_pad_fwd: bad geometry");
  const int64_t total = (int64_t)N * (H + 2 * pad) * (W + 2 * pad) * (cs / 4);
  replicate_pad_fwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, y, N, H, W, cs, pad);
  return cat::check_launch("replicate_pad_fwd");
}

int cat_replicate_pad_bwd(const float* dyp, float* dx, int N, int H, int W, int C, int cs, int pad, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 8.0 * N * (H + 2 * pad) * (W + 2 * pad) * cs, stream);
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && pad >= 0 && N > 0 && H > 0 && W > 0, "replicate_pad_bwd: bad geometry");
  const int64_t total = (int64_t)N * H * W * (cs / 4);
  replicate_pad_bwd_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(dyp, dx, N, H, W, cs, pad);
  return cat::check_launch("replicate_pad_bwd");
}

int cat_act_fwd(const float* x, float* y, int64_t n, int act, float slope, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 8.0 * n, stream);
  CAT_REQUIRE(n % 4 == 0, "act_fwd: n must be a multiple of 4");
  act_fwd_kernel<<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>(x, y, n / 4, act, slope);
  return cat::check_launch("act_fwd");
}

int cat_act_bwd(const float* y, const float* dy, float* dx, int64_t n, int act, float slope, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 12.0 * n, stream);
  CAT_REQUIRE(n % 4 == 0, "act_bwd: n must be a multiple of 4");
  act_bwd_kernel<<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>(y, dy, dx, n / 4, act, slope);
  return cat::check_launch("act_bwd");
}

int cat_add_n(const float* const* srcs, int nsrc, float* dst, int64_t n, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 4.0 * n * (nsrc + 1), stream);
  CAT_REQUIRE(nsrc >= 1 && nsrc <= 8 && n % 4 == 0, "add_n: 1..8 sources, n multiple of 4");
  AddSrcs s{};
  for (int i = 0; i < nsrc; ++i) s.p[i] = srcs[i];
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(n / 4);
  switch (nsrc) {
    case 1: add_n_kernel<1><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 2: add_n_kernel<2><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 3: add_n_kernel<3><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 4: add_n_kernel<4><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 5: add_n_kernel<5><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 6: add_n_kernel<6><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    case 7: add_n_kernel<7><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
    default: add_n_kernel<8><<<grid, 256, 0, st>>>(s, dst, n / 4); break;
  }
  return cat::check_launch("add_n");
}

int cat_concat2(const float* a, int ca, int acs, const float* b, int cb, int bcs, float* y, int ycs, int64_t M, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 8.0 * M * ycs, stream);
  CAT_REQUIRE(ca + cb <= ycs, "concat2: ycs too small");
  concat2_kernel<<<ew_grid(M * ycs), 256, 0, (hipStream_t)stream>>>(a, ca, acs, b, cb, bcs, y, ycs, M);
  return cat::check_launch("concat2");
}

int cat_slice_channels(const float* x, int xcs, int c0, int c, float* y, int ycs, int64_t M, cat_stream_t stream) {
  cat::ProfScope prof("elementwise", 0.0, 8.0 * M * ycs, stream);
  CAT_REQUIRE(c0 + c <= xcs && c <= ycs, "slice_channels: bad range");
  slice_kernel<<<ew_grid(M * ycs), 256, 0, (hipStream_t)stream>>>(x, xcs, c0, c, y, ycs, M);
  return cat::check_launch("slice_channels");
}

int cat_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ycs, cat_stream_t stream) {
  cat::ProfScope prof("layout", 0.0, 8.0 * N * C * H * W, stream);
  CAT_REQUIRE(ycs >= C, "nchw_to_nhwc: ycs < C");
  if (ycs == 4 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    const int64_t total = (int64_t)N * H * W;
    nchw_to_nhwc4_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, y, C, H * W, total);
    return cat::check_launch("nchw_to_nhwc");
  }
  dim3 grid(cdiv(H * W, 64), cdiv(ycs, 64), N);
  nchw_to_nhwc_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, C, H * W, ycs);
  return cat::check_launch("nchw_to_nhwc");
}

int cat_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int xcs, cat_stream_t stream) {
  cat::ProfScope prof("layout", 0.0, 8.0 * N * C * H * W, stream);
  CAT_REQUIRE(xcs >= C, "nhwc_to_nchw: xcs < C");
  dim3 grid(cdiv(H * W, 64), cdiv(C, 64), N);
  nhwc_to_nchw_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, C, H * W, xcs);
  return cat::check_launch("nhwc_to_nchw");
}

int cat_fill(float* p, int64_t n, float v, cat_stream_t stream) {
  cat::ProfScope prof("fill", 0.0, 4.0 * n, stream);
  if (n <= 0) return 0;
  fill_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(p, n, v);
  return cat::check_launch("fill");
}

int cat_axpy(float* y, const float* x, int64_t n, float a, cat_stream_t stream) {
  if (n <= 0) return 0;
  axpy_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(y, x, n, a);
  return cat::check_launch("axpy");
}

int cat_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, cat_stream_t stream) {
  cat::ProfScope prof("adam", 0.0, 28.0 * n, stream);
  CAT_REQUIRE(step >= 1, "adam: step is 1-based");
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  adam_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
                                                            grad_scale);
  return cat::check_launch("adam");
}

int cat_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper, float grad_scale, cat_stream_t stream) {
  cat::ProfScope prof("adam", 0.0, 28.0 * n, stream);
  CAT_REQUIRE(hyper, "adam: device hyper-parameter block required");
  hipStream_t s = (hipStream_t)stream;
  adam_tick_kernel<<<1, 1, 0, s>>>(hyper);
  if (n > 0) adam_dev_kernel<<<ew_grid(n), 256, 0, s>>>(p, g, m, v, n, hyper, grad_scale);
  return cat::check_launch("adam");
}

size_t cat_channel_sum_ws_bytes(int M, int cs) { return (size_t)cs_plan(M, cs).nb * cs * sizeof(float); }

int cat_channel_sum(const float* x, int M, int C, int cs, float* out, int accumulate, void* ws, cat_stream_t stream) {
  cat::ProfScope prof("channel_sum", 0.0, 4.0 * M * cs, stream);
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && ws, "channel_sum: bad arguments");
  const CsPlan p = cs_plan(M, cs);
  hipStream_t s = (hipStream_t)stream;
  chansum_partial_kernel<<<dim3(p.nb, p.nz), 256, 0, s>>>(x, (float*)ws, M, cs, p.zq, p.ppl, p.nb);
  chansum_final_kernel<<<cdiv(C, 16), 256, 0, s>>>((const float*)ws, out, C, cs, p.nb, accumulate);
  return cat::check_launch("channel_sum");
}

int cat_reflect_pad_bwd(const float* dxp, float* dx, int N, int H, int W, int C, int cs, int pad, cat_stream_t stream) {
  cat::ProfScope prof("reflect_pad_bwd", 0.0, 8.0 * N * H * W * cs, stream);
  CAT_REQUIRE(cs % 4 == 0 && cs >= C && pad < H && pad < W, "reflect_pad_bwd: bad geometry");
  const int64_t total = (int64_t)N * H * W * (cs / 4);
  reflect_fold_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(dxp, dx, N, H, W, cs, pad);
  return cat::check_launch("reflect_pad_bwd");
}

}  // extern "C"
