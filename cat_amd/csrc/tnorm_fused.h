// In-kernel finalize of a block stage's train-mode norms (gfx950): the LAST workgroup of the producing launch to leave its tile statistics
// behind merges the table and writes scale / shift / mean / rstd / running statistics -- what cat_tnorm_finalize did as a separate,
// dependent ~9 us launch (30 per student forward).  Nobody waits inside the launch (no grid barrier, no residency assumption, nothing to
// hang): a workgroup arrives on a counter and leaves; only the one that observes "I am last" does more work.
//
// Two-level merge in an order fixed by TILE INDEX, never by arrival order (replay / replica bit-identity):
//   tiles of a statistics group (the batch for BatchNorm, one image for InstanceNorm) are cut into sub-groups of SUB consecutive tiles;
//   the last arriver of a sub-group folds its tiles to (pixels, sum, M2) per channel (two passes: exact, no E[x^2] - E[x]^2);
//   the last sub-group merger folds those partials (Chan's formula) and finalises every norm module of the stage.
// Visibility (guide, guideline 16): per-XCD L2s are not coherent with each other and the per-CU L1 is never refreshed by other CUs' stores --
// the statistics and partials are written WRITE-THROUGH (agent-scope relaxed atomic stores), the storing wave drains them
// (s_waitcnt vmcnt(0)) before ONE lane bumps the counter, and the merging workgroup issues ONE agent-scope acquire before its plain loads.
// The counters return to zero inside the launch that used them (the top merger resets them after everyone has arrived).
#pragma once
#include "common.h"

namespace cat_fin {

constexpr int SUB = 32;      // tiles per sub-group

struct Dev {
  const float* gamma; const float* beta;      // [scs] or null
  float* scale; float* shift;                 // [G][scs]
  float* mean; float* rstd;                   // [G][mstride]
  unsigned* sync;                             // [G][1 + nsub]: top counter, sub-group counters (zero before the launch; zero after it)
  float* sub;                                 // [G][nsub][3][scs]
  int mstride, G, nsub, nslices;
  int ntile;                                  // tiles per statistics group
  int tiles_x, Ho, Wo, th, tw, per_img;       // tile lattice (pixels of tile t: from its position)
  float eps, momentum;
  cat_nslice_t sl[CAT_TNORM_MAXSLICE];
};

__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by EVERY thread of the workgroup after the workgroup's statistics rows were stored with st_wt() and each storing wave executed
// `asm volatile("s_waitcnt vmcnt(0)" ::: "memory")`; `several_waves`: the stores came from more than the wave thread 0 belongs to.
// tt = the tile's row in the table (rows of one statistics group are consecutive); part = the table [rows][2][scs]; flag = one LDS word.
__device__ __forceinline__ void arrive_and_finalize(const Dev& f, const float* part, int scs, int tt, int* flag, bool several_waves) {
  const int tid = threadIdx.x;
  const int g = f.G == 1 ? 0 : tt / f.ntile, ti = tt - g * f.ntile;
  const int s = ti / SUB;
  unsigned* sync = f.sync + (size_t)g * (1 + f.nsub);
  if (several_waves) __syncthreads();
  if (tid == 0) {
    const int mine = min(SUB, f.ntile - s * SUB);
    const unsigned old = __hip_atomic_fetch_add(sync + 1 + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == (unsigned)(mine - 1);
  }
  __syncthreads();
  if (!*flag) return;
  // ---- this workgroup merges sub-group s
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  const float tile_full = (float)(f.th * f.tw);
  const bool full = f.Ho % f.th == 0 && f.Wo % f.tw == 0;
  auto tile_n = [&](int t) {
    if (full) return tile_full;
    const int tl = t % f.per_img;
    const int ty = tl / f.tiles_x, tx = tl - ty * f.tiles_x;
    return (float)(min(f.th, f.Ho - ty * f.th) * min(f.tw, f.Wo - tx * f.tw));
  };
  float* subg = f.sub + (size_t)g * f.nsub * 3 * scs;
  {
    const int t0 = s * SUB, nt = min(SUB, f.ntile - t0);
    const float* pg = part + ((size_t)g * f.ntile + t0) * 2 * scs;
    for (int c = tid; c < scs; c += 256) {
      // two passes over the sub-group's rows (the second one hits L1): nothing but a few running values lives in registers -- this code
      // shares its kernel's register allocation with the MFMA loop and must not lower the kernel's occupancy
      const float* pc = pg + c;
      float S = 0.f, N = 0.f;
#pragma unroll 4
      for (int r = 0; r < nt; ++r) {
        S += pc[(size_t)r * 2 * scs];
        N += tile_n(t0 + r);
      }
      const float mean = S / N;
      float M2 = 0.f;
#pragma unroll 4
      for (int r = 0; r < nt; ++r) {
        const float n = tile_n(t0 + r), d = pc[(size_t)r * 2 * scs] / n - mean;
        M2 += pc[(size_t)r * 2 * scs + scs] + n * d * d;
      }
      st_wt(subg + ((size_t)s * 3 + 0) * scs + c, N);
      st_wt(subg + ((size_t)s * 3 + 1) * scs + c, S);
      st_wt(subg + ((size_t)s * 3 + 2) * scs + c, M2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == (unsigned)(f.nsub - 1);
  }
  __syncthreads();
  if (!*flag) return;
  // ---- this workgroup is the last merger: fold the sub-group partials and finalise the stage's norms
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  const int imgs = f.ntile / f.per_img;
  const float count = (float)f.Ho * (float)f.Wo * (float)imgs;
  for (int c = tid; c < scs; c += 256) {
    int sl = -1;
    for (int k = 0; k < f.nslices; ++k)
      if (c >= f.sl[k].c0 && c < f.sl[k].c0 + f.sl[k].c) sl = k;
    const int idx = g * scs + c, midx = g * f.mstride + c;
    if (sl < 0) {
      f.scale[idx] = f.shift[idx] = 0.f;
      if (c < f.mstride) f.mean[midx] = f.rstd[midx] = 0.f;
      continue;
    }
    float S = 0.f;
    for (int k = 0; k < f.nsub; ++k) S += subg[((size_t)k * 3 + 1) * scs + c];
    const float mean = S / count;
    float m2 = 0.f;
    for (int k = 0; k < f.nsub; ++k) {
      const float n = subg[((size_t)k * 3 + 0) * scs + c], d = subg[((size_t)k * 3 + 1) * scs + c] / n - mean;
      m2 += subg[((size_t)k * 3 + 2) * scs + c] + n * d * d;
    }
    float var = m2 / count;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + f.eps);
    f.mean[midx] = mean;
    f.rstd[midx] = rstd;
    const cat_nslice_t& SL = f.sl[sl];
    const int cl = c - SL.c0;
    if (SL.running_mean) {      // BatchNorm2d.train(): momentum update with the unbiased variance
      SL.running_mean[cl] = (1.f - f.momentum) * SL.running_mean[cl] + f.momentum * mean;
      const float unb = count > 1.f ? var * count / (count - 1.f) : var;
      SL.running_var[cl] = (1.f - f.momentum) * SL.running_var[cl] + f.momentum * unb;
      if (SL.num_batches && cl == 0) *SL.num_batches += 1;
    }
    const float ga = f.gamma ? f.gamma[c] : 1.f, be = f.beta ? f.beta[c] : 0.f;
    f.scale[idx] = ga * rstd;
    f.shift[idx] = be - mean * ga * rstd;
  }
  // everyone has arrived: return the counters to zero for the next launch that uses them
  for (int k = tid; k < 1 + f.nsub; k += 256) __hip_atomic_store(sync + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace cat_fin

// host side: cat_tfin_t (include/cat_hip.h) -> the by-value kernel argument
static inline int cat_fin_make(cat_fin::Dev& d, const cat_tfin_t* f, int scs, int N, int Ho, int Wo, int th, int tw) {
  if (f == nullptr) {
    d = cat_fin::Dev{};
    return 0;
  }
  CAT_REQUIRE(f->scale && f->shift && f->mean && f->rstd && f->sync && f->sub, "fused finalize: null buffer");
  CAT_REQUIRE(f->G == 1 || f->G == N, "fused finalize: groups must be 1 (batch norm) or N (instance norm)");
  CAT_REQUIRE(f->nslices >= 1 && f->nslices <= CAT_TNORM_MAXSLICE && (scs & 3) == 0, "fused finalize: %d slices", f->nslices);
  d.gamma = f->gamma; d.beta = f->beta; d.scale = f->scale; d.shift = f->shift; d.mean = f->mean; d.rstd = f->rstd;
  d.sync = f->sync; d.sub = f->sub; d.mstride = f->mstride; d.G = f->G; d.nslices = f->nslices; d.eps = f->eps; d.momentum = f->momentum;
  d.th = th; d.tw = tw; d.Ho = Ho; d.Wo = Wo;
  d.tiles_x = cat::cdiv(Wo, tw);
  d.per_img = d.tiles_x * cat::cdiv(Ho, th);
  d.ntile = d.per_img * (f->G == 1 ? N : 1);
  d.nsub = cat::cdiv(d.ntile, cat_fin::SUB);
  for (int k = 0; k < f->nslices; ++k) {
    d.sl[k] = f->slices[k];
    CAT_REQUIRE(f->slices[k].c0 >= 0 && f->slices[k].c > 0 && f->slices[k].c0 + f->slices[k].c <= scs, "fused finalize: slice %d outside the table", k);
  }
  return 0;
}
