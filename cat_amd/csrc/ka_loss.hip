// Kernel-alignment (KA) distillation loss and the scalar GAN / reconstruction losses, gfx950.
//
// KA(X, Y) = <Gx, Gy> / sqrt(<Gx,Gx> <Gy,Gy>),  Gx = X X^T (N x N over the batch), utils/common.py:38-46.
// The Gram is a pure HBM stream: N rows of D = C*H*W floats are read once.  Each wave owns a contiguous slice of D
// and accumulates the 16x16 Gram tiles with v_mfma_f32_16x16x4_f32 where BOTH operands are the same register
// (A[i][k] = B[k][i] = X[i][k]), so the matrix pipe does the N^2 work while the lanes only issue float4 loads:
// lane l loads X[row = l&15][k0 + 4*(l>>4) .. +3]  (4 lanes x 16 B = 64 contiguous bytes per row per load).
// Partials are written per workgroup and reduced by a single finalize block (deterministic).
// Backward: dX = gout * 2 * Cx X with the N x N matrix Cx = Gy/sqrt(sxx syy) - sxy Gx/(sxx^1.5 syy^0.5).
#include "common.h"

namespace {
using cat::cdiv;

constexpr int KA_MAXN = 64;

template <int NT>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ X, int64_t D, int N, float* __restrict__ part,
                                                   int64_t chunk) {
  // part: [gridDim.x][NT*16][NT*16]
  __shared__ float red[4][NT * NT][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t kbeg = ((int64_t)blockIdx.x * 4 + wave) * chunk;
  const int64_t kend = kbeg + chunk < D ? kbeg + chunk : D;
  f4 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const float* rowp[NT];
  bool rv[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int r = i * 16 + lr;
    rv[i] = r < N;
    rowp[i] = X + (int64_t)(rv[i] ? r : 0) * D + lq * 4;
  }
  for (int64_t k = kbeg; k < kend; k += 16) {
    f4 v[NT];
    const bool kv = k + lq * 4 < kend;  // chunk and D are multiples of 4
#pragma unroll
    for (int i = 0; i < NT; ++i) v[i] = (rv[i] && kv) ? *reinterpret_cast<const f4*>(rowp[i] + k) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i][t], v[j][t], acc[i][j], 0, 0, 0);
  }
  // D[row = lq*4 + rg][col = lr] per tile -> LDS, then sum the 4 waves
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) red[wave][i * NT + j][(lq * 4 + rg) * 16 + lr] = acc[i][j][rg];
  __syncthreads();
  constexpr int NN = NT * 16;
  float* dst = part + (int64_t)blockIdx.x * NN * NN;
  for (int e = tid; e < NT * NT * 256; e += 256) {
    const int tile = e >> 8, o = e & 255;
    const float s = red[0][tile][o] + red[1][tile][o] + red[2][tile][o] + red[3][tile][o];
    const int ti = tile / NT, tj = tile - ti * NT;
    dst[(ti * 16 + (o >> 4)) * NN + tj * 16 + (o & 15)] = s;
  }
}

// ws layout (floats): [0..3] sxy, sxx, syy, ka | Gx[N*N] | Gy[N*N] | partX[nb][NN*NN] | partY[nb][NN*NN]
// stage 1: Gx / Gy entries = sum of the per-workgroup partials; 16 entries x 16 partial-lanes per workgroup, four independent loads per lane and
// step, lanes combined pairwise through LDS in a fixed order (round 6: 64 entries x 4 lanes = 4 workgroups walked 2 x 128 dependent loads
// per lane: 43 us per loss term for 2 MB of partials)
__global__ __launch_bounds__(256) void ka_gram_reduce_kernel(float* __restrict__ ws, int N, int NN, int nbx, int nby) {
  __shared__ float red[2][256];
  float* Gx = ws + 4;
  float* Gy = Gx + N * N;
  const float* px = Gy + N * N;
  const float* py = px + (int64_t)nbx * NN * NN;
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), zl = threadIdx.x >> 4;
  float gx = 0.f, gy = 0.f;
  if (e < N * N) {
    const int i = e / N, j = e - i * N;
    const int64_t zs = (int64_t)NN * NN;
    const float* sx = px + i * NN + j;
    const float* sy = py + i * NN + j;
    int b = zl;
    for (; b + 48 < nbx; b += 64) {
      const float v0 = sx[b * zs], v1 = sx[(b + 16) * zs], v2 = sx[(b + 32) * zs], v3 = sx[(b + 48) * zs];
      gx += v0;
      gx += v1;
      gx += v2;
      gx += v3;
    }
    for (; b < nbx; b += 16) gx += sx[b * zs];
    b = zl;
    for (; b + 48 < nby; b += 64) {
      const float v0 = sy[b * zs], v1 = sy[(b + 16) * zs], v2 = sy[(b + 32) * zs], v3 = sy[(b + 48) * zs];
      gy += v0;
      gy += v1;
      gy += v2;
      gy += v3;
    }
    for (; b < nby; b += 16) gy += sy[b * zs];
  }
  red[0][threadIdx.x] = gx;
  red[1][threadIdx.x] = gy;
  __syncthreads();
#pragma unroll
  for (int st = 8; st >= 1; st >>= 1) {
    if (zl < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st * 16];
      red[1][threadIdx.x] += red[1][threadIdx.x + st * 16];
    }
    __syncthreads();
  }
  if (zl == 0 && e < N * N) {
    Gx[e] = red[0][threadIdx.x];
    Gy[e] = red[1][threadIdx.x];
  }
}

// stage 2: ws[0..3] = sxy, sxx, syy, ka
__global__ __launch_bounds__(256) void ka_finalize_kernel(float* __restrict__ ws, int N, float* __restrict__ out) {
  __shared__ float red[3][256];
  const float* Gx = ws + 4;
  const float* Gy = Gx + N * N;
  float sxy = 0.f, sxx = 0.f, syy = 0.f;
  for (int e = threadIdx.x; e < N * N; e += 256) {
    const float gx = Gx[e], gy = Gy[e];
    sxy += gx * gy;
    sxx += gx * gx;
    syy += gy * gy;
  }
  red[0][threadIdx.x] = sxy;
  red[1][threadIdx.x] = sxx;
  red[2][threadIdx.x] = syy;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float a = red[0][0], b = red[1][0], c = red[2][0];
    const float ka = a / sqrtf(b * c);
    ws[0] = a;
    ws[1] = b;
    ws[2] = c;
    ws[3] = ka;
    out[0] = ka;
  }
}

// dX[i][k] = gout * 2 * sum_j Cx[i][j] X[j][k]; one lane per float4 column, 8 output rows at a time.
__global__ __launch_bounds__(256) void ka_bwd_kernel(const float* __restrict__ X, int64_t D, int N, const float* __restrict__ gout,
                                                     const float* __restrict__ ws, float* __restrict__ dX) {
  __shared__ float C[KA_MAXN * KA_MAXN];
  const float sxy = ws[0], sxx = ws[1], syy = ws[2];
  const float* Gx = ws + 4;
  const float* Gy = Gx + N * N;
  const float g = gout[0] * 2.f;
  const float ca = g / sqrtf(sxx * syy), cb = g * sxy / (sxx * sqrtf(sxx * syy));
  for (int e = threadIdx.x; e < N * N; e += 256) C[e] = ca * Gy[e] - cb * Gx[e];
  __syncthreads();
  const int64_t nq = D / 4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
    for (int i0 = 0; i0 < N; i0 += 8) {
      f4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < N; ++j) {
        const f4 xv = *reinterpret_cast<const f4*>(X + (int64_t)j * D + q * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float c = i0 + i < N ? C[(i0 + i) * N + j] : 0.f;
          acc[i] += xv * c;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i0 + i < N) *reinterpret_cast<f4*>(dX + (int64_t)(i0 + i) * D + q * 4) = acc[i];
    }
  }
}

struct GramPlan { int NT, NN, nb; int64_t chunk; };
GramPlan gram_plan(int N, int64_t D) {
  GramPlan p;
  p.NT = cdiv(N, 16);
  p.NN = p.NT * 16;
  int64_t nb = D / (4 * 512);  // >= 512 floats of every row per wave
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  p.nb = (int)nb;
  p.chunk = ((D + (int64_t)p.nb * 4 - 1) / ((int64_t)p.nb * 4) + 15) / 16 * 16;
  return p;
}
constexpr int KA_MAXNB = 512;

int launch_gram(const float* X, int64_t D, int N, float* part, const GramPlan& p, hipStream_t s) {
  switch (p.NT) {
    case 1: gram_kernel<1><<<p.nb, 256, 0, s>>>(X, D, N, part, p.chunk); break;
    case 2: gram_kernel<2><<<p.nb, 256, 0, s>>>(X, D, N, part, p.chunk); break;
    case 3: gram_kernel<3><<<p.nb, 256, 0, s>>>(X, D, N, part, p.chunk); break;
    default: gram_kernel<4><<<p.nb, 256, 0, s>>>(X, D, N, part, p.chunk); break;
  }
  return cat::check_launch("ka_gram");
}

// ------------------------------------------------------------------------------------------------ scalar losses
__device__ __forceinline__ float loss_term(int kind, float a, float b, float t) {
  switch (kind) {
    case 0: return fabsf(a - b);
    case 1: return (a - t) * (a - t);
    case 2: return -fminf(a - 1.f, 0.f);
    case 3: return -fminf(-a - 1.f, 0.f);
    case 4: return -a;
    case 6: return (1.f - t) * a + fmaxf(-a, 0.f) + log1pf(expf(-fabsf(a)));   // BCE with logits against the constant target t
    case 7: return a;
    default: return (a - b) * (a - b);
  }
}
__device__ __forceinline__ float loss_grad(int kind, float a, float b, float t) {
  switch (kind) {
    case 0: return a > b ? 1.f : (a < b ? -1.f : 0.f);
    case 1: return 2.f * (a - t);
    case 2: return a - 1.f < 0.f ? -1.f : 0.f;   // torch.min(x-1, 0): ties send the gradient to ... see tests (measure zero)
    case 3: return -a - 1.f < 0.f ? 1.f : 0.f;
    case 4: return -1.f;
    case 6: return 1.f / (1.f + expf(-a)) - t;   // sigmoid(a) - t
    case 7: return 1.f;
    default: return 2.f * (a - b);
  }
}

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, float t,
                                                           int64_t nquads, int nq, int C, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const f4 av = *reinterpret_cast<const f4*>(a + i * 4);
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b) bv = *reinterpret_cast<const f4*>(b + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) s += loss_term(kind, av[e], bv[e], t);
  }
  s = cat::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int nb, float inv_count, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = cat::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * inv_count;
}
__global__ __launch_bounds__(256) void loss_bwd_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, float t,
                                                       int64_t nquads, int nq, int C, const float* __restrict__ gout, float scale,
                                                       float* __restrict__ da) {
  const float g = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nq) * 4;
    const f4 av = *reinterpret_cast<const f4*>(a + i * 4);
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (b) bv = *reinterpret_cast<const f4*>(b + i * 4);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = c + e < C ? g * loss_grad(kind, av[e], bv[e], t) : 0.f;
    *reinterpret_cast<f4*>(da + i * 4) = o;
  }
}

int loss_nb(int64_t nquads) {
  int64_t b = (nquads + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" {

size_t cat_ka_ws_bytes(int N) {
  const int NN = cdiv(N, 16) * 16;
  return (size_t)(4 + 2 * N * N + 2 * (size_t)KA_MAXNB * NN * NN) * sizeof(float);
}

int cat_ka_fwd(const float* X, int64_t Dx, const float* Y, int64_t Dy, int N, float* out, void* ws, cat_stream_t stream) {
  CAT_REQUIRE(N >= 1 && N <= KA_MAXN, "ka: batch N=%d outside [1,%d]", N, KA_MAXN);
  CAT_REQUIRE(Dx % 4 == 0 && Dy % 4 == 0 && ws, "ka: row lengths must be multiples of 4");
  const GramPlan px = gram_plan(N, Dx), py = gram_plan(N, Dy);
  cat::ProfScope prof("ka_fwd", 2.0 * N * N * (double)(Dx + Dy), 4.0 * N * (double)(Dx + Dy), stream);
  float* w = (float*)ws;
  float* partx = w + 4 + 2 * N * N;
  float* party = partx + (int64_t)px.nb * px.NN * px.NN;
  hipStream_t s = (hipStream_t)stream;
  if (int e = launch_gram(X, Dx, N, partx, px, s)) return e;
  if (int e = launch_gram(Y, Dy, N, party, py, s)) return e;
  ka_gram_reduce_kernel<<<cdiv(N * N, 16), 256, 0, s>>>(w, N, px.NN, px.nb, py.nb);
  ka_finalize_kernel<<<1, 256, 0, s>>>(w, N, out);
  return cat::check_launch("ka_finalize");
}

int cat_ka_bwd(const float* X, int64_t Dx, int N, const float* gout, const void* ws, float* dX, cat_stream_t stream) {
  CAT_REQUIRE(N >= 1 && N <= KA_MAXN && Dx % 4 == 0, "ka bwd: bad arguments");
  int64_t nb = (Dx / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  cat::ProfScope prof("ka_bwd", 2.0 * N * N * (double)Dx, 8.0 * N * (double)Dx, stream);
  ka_bwd_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(X, Dx, N, gout, (const float*)ws, dX);
  return cat::check_launch("ka_bwd");
}

size_t cat_loss_ws_bytes(int64_t M) { (void)M; return 1024 * sizeof(float); }

int cat_loss_fwd(int kind, const float* a, const float* b, float target, int64_t M, int C, int cs, float* out, void* ws,
                 cat_stream_t stream) {
  CAT_REQUIRE(kind >= 0 && kind <= 7 && cs % 4 == 0 && cs >= C && ws, "loss: bad arguments");
  CAT_REQUIRE((kind != 0 && kind != 5) || b, "loss: kind %d needs a second tensor", kind);
  const int64_t nquads = M * (cs / 4);
  const int nb = loss_nb(nquads);
  cat::ProfScope prof("loss", 0.0, 4.0 * M * cs * (b ? 2 : 1), stream);
  hipStream_t s = (hipStream_t)stream;
  loss_partial_kernel<<<nb, 256, 0, s>>>(kind, a, b, target, nquads, cs / 4, C, (float*)ws);
  loss_final_kernel<<<1, 256, 0, s>>>((const float*)ws, nb, 1.f / (float)((double)M * C), out);
  return cat::check_launch("loss_fwd");
}

int cat_loss_bwd(int kind, const float* a, const float* b, float target, int64_t M, int C, int cs, const float* gout, float scale,
                 float* da, cat_stream_t stream) {
  CAT_REQUIRE(kind >= 0 && kind <= 7 && cs % 4 == 0 && cs >= C, "loss: bad arguments");
  const int64_t nquads = M * (cs / 4);
  int64_t nb = (nquads + 255) / 256;
  if (nb > 8192) nb = 8192;
  cat::ProfScope prof("loss", 0.0, 4.0 * M * cs * (b ? 3 : 2), stream);
  loss_bwd_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(kind, a, b, target, nquads, cs / 4, C, gout, scale / (float)((double)M * C), da);
  return cat::check_launch("loss_bwd");
}

}  // extern "C"
