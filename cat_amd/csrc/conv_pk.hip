// LDS-tile convolution with pre-packed filter fragments ("tconv"), stride 1, k in {1, 3, 5}, NHWC fp32, gfx950.
//
//   out[n][oy][ox][co] = act( bias[co] + sum over K SEGMENTS s, taps (ky, kx) of s, channels c of s
//                                         f_s(src_s[n][oy - padv_s + ky][ox - padv_s + kx][c]) * W_s(tap, c, co) )
//
// One kernel body serves
//   * a single "same" convolution (one segment): the pruned student's ragged 77 -> 7..23 and 7..23 -> 77 layers and the
//     teacher's 256 -> 42 -> 256 pairs (models/modules/inception_modules.py:135-147), whose im2col kernels re-fetch the A operand
//     from L2 once per tap (13x over-fetch measured on 256 -> 42 5x5);
//   * its input gradient (segment source = dy, filters packed transposed + flipped, padv = k - 1 - pad);
//   * the K-CONCATENATED sum of several convolutions of different kernel sizes reading different tensors -- the branch sum of
//     InvertedResidualChannels.forward (:230-236: sum_k res_k(x) + sum_k dw_k(x) as ONE launch writing the sum once), and the sum
//     of the six first-conv input gradients in the backward pass;
//   * f_s = optional per-channel affine + activation applied while the tile is staged (a train-mode norm's apply pass folded
//     into the consumer: no normalised copy of the hidden tensor is written to HBM).
//
// Workgroup = 256 threads = 4 wave64 = 8 x TW output pixels (TW = 16 or 32); wave w owns rows 2w, 2w+1 as MT = TW/8 M-tiles of
// 16 consecutive pixels.  Per 16-channel chunk the (8 + halo) x (TW + halo) source patch is staged ONCE in LDS as
// [row][col][16 ch + 4 pad] (double buffered, one barrier per chunk); the K index inside a chunk runs over (tap, channel quad)
// PAIRS, four pairs per v_mfma_f32_16x16x4_f32 group (lane quarter lq <-> pair 4g + lq), so ragged channel counts cost at most
// one partially filled group per chunk instead of a zero-padded 16-channel chunk per tap (18 hidden channels: K efficiency 0.56
// -> 0.98).  The A fragment of (M-tile, group) is one ds_read_b128 at a per-lane offset read from a small LDS table; the B
// fragments come from a stream that cat_tconv_pack laid out in exactly the order the kernel consumes it ([chunk][group][N-tile]
// [lane][4]: one coalesced 1 KB global_load_dwordx4 per wave, no address arithmetic on the vector ALU, which on gfx950 shares
// its issue slots with the fp32 MFMA).  Filters of trainable layers are re-packed once per optimizer step.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace cat_pk {

constexpr int TH = 8, PITCH = 20, TABN = 128;

__device__ __attribute__((aligned(16))) float g_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct Launch {
  int hl, tr, tc;          // halo (left / top), staged rows / columns
  int tiles_x, tiles;      // output tiles per row / per image
  int nt_total, nblk;      // 16-wide N tiles of the output, N blocks (gridDim.x = N * tiles * nblk)
  int ablate;              // DIAGNOSTIC BUILD ONLY (cat::kDiag; CAT_PK_ABLATE, results become wrong): 1 every group reads the SAME filter block
                           // (L1 hits), 2 every A fragment reads LDS offset 0, 4 staging loads hit one cached line, 8 no staging / barrier after
                           // the first chunk.  The production kernels read `abl`, a compile-time 0
};

__device__ __forceinline__ f4 bload(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff_) {
  const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff_, 0);
  static_assert(sizeof(r) == sizeof(f4), "buffer load width");
  return __builtin_bit_cast(f4, r);
}

// MFMA stream of one staged chunk: ngr groups of 4 (tap, quad) pairs, MT x NT accumulator tiles.  Two operand register sets in
// ping-pong (no copies); the operands of group g+1 and the A offset of group g+2 are requested BEFORE the 4*MT*NT MFMAs of group g are
// issued, so no load is waited for in the group that issued it.  tab = this lane quarter's A-offset table, so = byte offset of the
// chunk's first group in the packed filter stream, gbytes = bytes per group.
template <int MT, int NT>
__device__ __forceinline__ void mma_groups(f4 (&acc)[MT][NT], const float* tile, const int* tab, const int (&abase)[MT],
                                           const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&jb)[NT], unsigned so, unsigned gbytes, int ngr) {
  const int last = ngr - 1;
  f4 a0[MT], a1[MT], b0[NT], b1[NT];
  int off = tab[0];
  int offn = tab[4 * min(1, last)];
#pragma unroll
  for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const f4*>(tile + abase[i] + off);
#pragma unroll
  for (int j = 0; j < NT; ++j) b0[j] = bload(rsrc, jb[j], so);
  int gi = 0;
  while (true) {
    {
      const int g1 = min(gi + 1, last), g2 = min(gi + 2, last);
#pragma unroll
      for (int i = 0; i < MT; ++i) a1[i] = *reinterpret_cast<const f4*>(tile + abase[i] + offn);
#pragma unroll
      for (int j = 0; j < NT; ++j) b1[j] = bload(rsrc, jb[j], so + (unsigned)g1 * gbytes);
      offn = tab[4 * g2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][tq], b0[j][tq], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++gi >= ngr) break;
    {
      const int g1 = min(gi + 1, last), g2 = min(gi + 2, last);
#pragma unroll
      for (int i = 0; i < MT; ++i) a0[i] = *reinterpret_cast<const f4*>(tile + abase[i] + offn);
#pragma unroll
      for (int j = 0; j < NT; ++j) b0[j] = bload(rsrc, jb[j], so + (unsigned)g1 * gbytes);
      offn = tab[4 * g2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][tq], b1[j][tq], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++gi >= ngr) break;
  }
}

template <int NT, int TW>
__global__ __launch_bounds__(256) void tconv_kernel(const cat_tconv_t g, const float* __restrict__ pack, const float* __restrict__ bias,
                                                    float* __restrict__ y, const Launch L) {
  constexpr int MT = TW / 8;
  constexpr int MAXIT = ((TH + 4) * (TW + 4) * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_floats = L.tr * L.tc * PITCH;
  float* tile0 = smem;
  int* tab0 = reinterpret_cast<int*>(smem + 2 * tile_floats);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  // block -> (image, tile, N block): N blocks of one tile are adjacent (they stage the same patch), tiles of one image adjacent
  const int bid = cat::xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % L.nblk, tt = bid / L.nblk;
  const int n = tt / L.tiles, t = tt - n * L.tiles;
  const int oy0 = (t / L.tiles_x) * TH, ox0 = (t % L.tiles_x) * TW;
  const int j0 = nb * NT;
  const int slots = L.tr * L.tc * 4;
  const int quad = tid & 3;
  const int abl = cat::kDiag ? L.ablate : 0;

  // staging map: this thread's patch pixels (independent of the segment)
  int sy[MAXIT], sx[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int pix = (tid + it * 256) >> 2;
    const int r = pix / L.tc;
    sy[it] = oy0 - L.hl + r;
    sx[it] = ox0 - L.hl + (pix - r * L.tc);
  }

  // cursor over (segment, 16-channel chunk)
  int cs_ = 0, cc0 = 0;                 // segment, first channel of the chunk
  int64_t cpack = g.seg[0].pack_off;    // float offset of the chunk's packed filters
  unsigned soff[MAXIT];
  unsigned smask = 0;
  auto locate = [&](int s) {   // source offsets of the patch pixels for segment s
    const int refl = g.seg[s].reflect, xcs = g.seg[s].xcs;
    smask = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int iy = sy[it], ix = sx[it];
      bool v = tid + it * 256 < slots;
      if (refl) {
        v = v && iy > -g.H && iy < 2 * g.H - 1 && ix > -g.W && ix < 2 * g.W - 1;
        iy = cat::reflect_idx(iy, g.H);
        ix = cat::reflect_idx(ix, g.W);
      } else {
        v = v && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
      }
      smask |= v ? (1u << it) : 0u;
      soff[it] = v ? ((unsigned)(n * g.H + iy) * (unsigned)g.W + (unsigned)ix) * (unsigned)xcs + quad * 4 : 0u;   // < 2^32 elements (host-checked)
    }
    if (abl & 4) smask = 0;
  };
  f4 sreg[MAXIT], ssc, ssh;
  int s_act = 0;
  float s_neg = 1.f;
  bool s_aff = false, s_qv = false;
  auto gload = [&](int s, int c0) {
    const float* src = g.seg[s].src;
    s_qv = c0 + quad * 4 < g.seg[s].c4;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const bool v = ((smask >> it) & 1u) && s_qv;
      sreg[it] = *reinterpret_cast<const f4*>(v ? src + soff[it] + c0 : g_zero);
    }
    s_aff = g.seg[s].scale != nullptr;
    s_act = g.seg[s].act;
    s_neg = s_act == CAT_ACT_RELU ? 0.f : (s_act == CAT_ACT_LRELU ? g.seg[s].slope : 1.f);
    if (s_aff) {
      const int so = n * g.seg[s].sstride + c0 + quad * 4;
      ssc = *reinterpret_cast<const f4*>(s_qv ? g.seg[s].scale + so : g_zero);
      ssh = *reinterpret_cast<const f4*>(s_qv ? g.seg[s].shift + so : g_zero);
    }
  };
  auto sstore = [&](int buf, int s, int c0) {
    float* tile = tile0 + buf * tile_floats;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = tid + it * 256;
      f4 v = sreg[it];
      if (s_aff || s_act) {   // wave-uniform
        const bool ok = ((smask >> it) & 1u) && s_qv;   // padding pixels / channels stay exactly 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = s_aff ? fmaf(v[e], ssc[e], ssh[e]) : v[e];
          a = a > 0.f ? a : a * s_neg;      // s_neg: 0 (ReLU), slope (LeakyReLU), 1 (none)
          v[e] = ok ? a : 0.f;
        }
      }
      if (idx < slots) *reinterpret_cast<f4*>(tile + (idx >> 2) * PITCH + quad * 4) = v;
    }
    // per-lane-quarter A offsets of every MFMA group of this chunk: pair p = 4 g + lq -> (tap, channel quad)
    const int ks = g.seg[s].ks, taps = ks * ks;
    const int nq = min(4, (g.seg[s].c4 - c0) >> 2);
    const int ngr = (taps * nq + 3) >> 2;
    if (tid < TABN) {
      int off = 0;
      if (tid < ngr * 4) {
        const int tap = tid / nq, qd = tid - tap * nq;
        if (tap < taps) {
          const int ky = tap / ks, kx = tap - ky * ks;
          const int d = L.hl - g.seg[s].padv;
          off = ((d + ky) * L.tc + d + kx) * PITCH + qd * 4;
        }
      }
      tab0[buf * TABN + tid] = (abl & 2) ? 0 : off;
    }
  };

  f4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = 2 * wave + (MT == 2 ? i : (i >> 1)), col = (MT == 2 ? 0 : (i & 1) * 16) + lr;
    abase[i] = (row * L.tc + col) * PITCH;
  }
  // B stream of this lane: [group][N tile][lane][4]; tiles beyond the output's last one re-read the last valid tile (masked at the store)
  // packed filters through a buffer resource: per-lane byte offset in voffset, the group's offset in soffset (scalar) -- no vector
  // ALU address arithmetic in the MFMA stream
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pack), 0, 0x7fffffff, 0x00020000);
  unsigned jb[NT];   // byte offsets inside one group
#pragma unroll
  for (int j = 0; j < NT; ++j) jb[j] = (unsigned)(min(j0 + j, L.nt_total - 1) * 64 + lane) * 16u;
  const int gstride = L.nt_total * 256;

  locate(0);
  gload(0, 0);
  sstore(0, 0, 0);
  __syncthreads();
  int buf = 0;
  while (true) {
    // next chunk
    int ns = cs_, nc0 = cc0 + 16;
    const int ks = g.seg[cs_].ks, taps = ks * ks;
    const int nq = min(4, (g.seg[cs_].c4 - cc0) >> 2);
    const int ngr = (taps * nq + 3) >> 2;
    int64_t npack = cpack + (int64_t)ngr * gstride;
    if (nc0 >= g.seg[cs_].c4) {
      ++ns;
      nc0 = 0;
      if (ns < g.nseg) {
        npack = g.seg[ns].pack_off;
        locate(ns);
      }
    }
    const bool more = ns < g.nseg;
    const bool stage = !(abl & 8);
    if (more && stage) gload(ns, nc0);   // in flight behind this chunk's MFMA stream

    mma_groups<MT, NT>(acc, tile0 + buf * tile_floats, tab0 + buf * TABN + lq, abase, prsrc, jb, (abl & 1) ? 0u : (unsigned)cpack * 4u,
                       (abl & 1) ? 0u : (unsigned)gstride * 4u, ngr);
    if (!more) break;
    if (stage) {
      sstore(buf ^ 1, ns, nc0);
      __syncthreads();
      buf ^= 1;
    }
    cs_ = ns;
    cc0 = nc0;
    cpack = npack;
  }

  if (g.stats) {
    // per-tile sum and sum of squared deviations from the TILE mean of the pre-activation output (two passes over the accumulators:
    // no E[x^2] - E[x]^2 cancellation); cat_tnorm_finalize merges the tiles with the exact pairwise formula
    float* red = smem + 2 * tile_floats + 2 * TABN;   // [2][4 waves][NT * 16]
    const int cnt = min(TH, g.Ho - oy0) * min(TW, g.Wo - ox0);
    float s[NT], mean[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = (j0 + j) * 16 + lr;
      const float b = (bias && co < g.Nn) ? bias[co] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bool rowv = oy0 + 2 * wave + (MT == 2 ? i : (i >> 1)) < g.Ho;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const bool v = rowv && ox0 + (MT == 2 ? 0 : (i & 1) * 16) + lq * 4 + rg < g.Wo;
          a += v ? acc[i][j][rg] + b : 0.f;
        }
      }
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      s[j] = a;
      if (lq == 0) red[wave * NT * 16 + j * 16 + lr] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = j * 16 + lr;
      s[j] = (red[c] + red[NT * 16 + c]) + (red[2 * NT * 16 + c] + red[3 * NT * 16 + c]);
      mean[j] = s[j] / (float)cnt;
    }
    float* red2 = red + 4 * NT * 16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = (j0 + j) * 16 + lr;
      const float b = (bias && co < g.Nn) ? bias[co] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bool rowv = oy0 + 2 * wave + (MT == 2 ? i : (i >> 1)) < g.Ho;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const bool v = rowv && ox0 + (MT == 2 ? 0 : (i & 1) * 16) + lq * 4 + rg < g.Wo;
          const float d = acc[i][j][rg] + b - mean[j];
          a += v ? d * d : 0.f;
        }
      }
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (lq == 0) red2[wave * NT * 16 + j * 16 + lr] = a;
    }
    __syncthreads();
    if (wave == 0 && lq == 0) {
      float* dst = g.stats + (int64_t)tt * 2 * g.scs;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = j * 16 + lr, co = (j0 + j) * 16 + lr;
        if (j0 + j < L.nt_total && co < g.ycw) {
          const bool cv = co < g.Nn;
          dst[co] = cv ? s[j] : 0.f;
          dst[g.scs + co] = cv ? (red2[c] + red2[NT * 16 + c]) + (red2[2 * NT * 16 + c] + red2[3 * NT * 16 + c]) : 0.f;
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int oy = oy0 + 2 * wave + (MT == 2 ? i : (i >> 1));
    if (oy >= g.Ho) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ox = ox0 + (MT == 2 ? 0 : (i & 1) * 16) + lq * 4 + rg;
      if (ox >= g.Wo) continue;
      float* yo = y + (((int64_t)n * g.Ho + oy) * g.Wo + ox) * g.ycs;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = (j0 + j) * 16 + lr;
        if (j0 + j >= L.nt_total) continue;
        if (co < g.Nn) {
          float v = cat::apply_act(acc[i][j][rg] + (bias ? bias[co] : 0.f), g.act, g.slope);
          if (g.res) v += g.res[(((int64_t)n * g.Ho + oy) * g.Wo + ox) * g.rcs + co];
          yo[co] = v;
        } else if (co < g.ycw) {
          yo[co] = 0.f;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused first convs of a block
// All first convs of an InvertedResidualChannels block (inception_modules.py:135-147,150-165: the 5x5, the 3x3 and the N-concatenated
// 1x1 convs of every branch) from ONE staging of the input tile per 16-channel chunk: three accumulator sets (NA / NB / NC 16-wide
// tiles), three filter streams, outputs = three channel slices of the pre-norm buffer + their per-tile statistics.  The narrow convs
// alone cannot hide their own staging latency (a 3x3 chunk is ~1 us of MFMA work); together a chunk carries ~7 us.
struct S1Args {
  const float* x; const float* pack[3]; const float* bias; float* ys[3]; float* stats;   // ys / ycs3: output buffer and pixel stride per slot
  int xcs, c4, N, H, W, reflect, ycs3[3], scs;
  int nt_total[3], col0[3], width[3], nvalid[3];
  int hl, tr, tc, tiles_x, tiles;
};

template <int NT, bool STATS>
__device__ __forceinline__ void s1_epilogue(f4 (&acc)[2][NT], const S1Args& p, int k, int n, int tt, int oy0, int ox0, int wave, int lr, int lq,
                                            float* red) {
  const int col0 = p.col0[k], width = p.width[k], nv = width;   // padding columns inside the slice carry zero filters: plain zeros come out
  const int cnt = min(TH, p.H - oy0) * min(16, p.W - ox0);
  float s[NT], mean[NT], bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = j * 16 + lr;
    bv[j] = (p.bias && co < nv) ? p.bias[col0 + co] : 0.f;
  }
  if constexpr (STATS) {     // per-tile statistics; the input-gradient use of the kernel has none
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool rowv = oy0 + 2 * wave + i < p.H;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) a += (rowv && ox0 + lq * 4 + rg < p.W) ? acc[i][j][rg] + bv[j] : 0.f;
      }
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (lq == 0) red[wave * NT * 16 + j * 16 + lr] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = j * 16 + lr;
      s[j] = (red[c] + red[NT * 16 + c]) + (red[2 * NT * 16 + c] + red[3 * NT * 16 + c]);
      mean[j] = s[j] / (float)cnt;
    }
    float* red2 = red + 4 * NT * 16;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool rowv = oy0 + 2 * wave + i < p.H;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float d = acc[i][j][rg] + bv[j] - mean[j];
          a += (rowv && ox0 + lq * 4 + rg < p.W) ? d * d : 0.f;
        }
      }
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (lq == 0) red2[wave * NT * 16 + j * 16 + lr] = a;
    }
    __syncthreads();
    if (wave == 0 && lq == 0) {
      float* dst = p.stats + (int64_t)tt * 2 * p.scs + col0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = j * 16 + lr;
        if (c < width) {
          const bool cv = c < nv;
          dst[c] = cv ? s[j] : 0.f;
          dst[p.scs + c] = cv ? (red2[c] + red2[NT * 16 + c]) + (red2[2 * NT * 16 + c] + red2[3 * NT * 16 + c]) : 0.f;
        }
      }
    }
    __syncthreads();   // red is reused by the next sub-convolution
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = oy0 + 2 * wave + i;
    if (oy >= p.H) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ox = ox0 + lq * 4 + rg;
      if (ox >= p.W) continue;
      float* yo = p.ys[k] + (((int64_t)n * p.H + oy) * p.W + ox) * p.ycs3[k] + col0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = j * 16 + lr;
        if (co < nv) yo[co] = acc[i][j][rg] + bv[j];
        else if (co < width) yo[co] = 0.f;
      }
    }
  }
}

template <int NA, int NB, int NC, bool STATS>
__global__ __launch_bounds__(256) void tstage1_kernel(const S1Args p) {
  constexpr int MT = 2, TW = 16, MAXIT = ((TH + 4) * (TW + 4) * 4 + 255) / 256;
  constexpr int NMAX = NA > NB ? (NA > NC ? NA : NC) : (NB > NC ? NB : NC);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_floats = p.tr * p.tc * PITCH;
  float* tile0 = smem;
  int* tab0 = reinterpret_cast<int*>(smem + 2 * tile_floats);      // [buf][3][TABN]
  float* red = smem + 2 * tile_floats + 6 * TABN;                  // [2][4][NMAX * 16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int tt = cat::xcd_remap(blockIdx.x, gridDim.x);
  const int n = tt / p.tiles, t = tt - n * p.tiles;
  const int oy0 = (t / p.tiles_x) * TH, ox0 = (t % p.tiles_x) * TW;
  const int slots = p.tr * p.tc * 4, quad = tid & 3;
  unsigned soff[MAXIT], smask = 0;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int pix = (tid + it * 256) >> 2;
    const int r = pix / p.tc;
    int iy = oy0 - p.hl + r, ix = ox0 - p.hl + (pix - r * p.tc);
    bool v = tid + it * 256 < slots;
    if (p.reflect) {
      v = v && iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
      iy = cat::reflect_idx(iy, p.H);
      ix = cat::reflect_idx(ix, p.W);
    } else {
      v = v && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    }
    smask |= v ? (1u << it) : 0u;
    soff[it] = v ? ((unsigned)(n * p.H + iy) * (unsigned)p.W + (unsigned)ix) * (unsigned)p.xcs + quad * 4 : 0u;
  }
  f4 sreg[MAXIT];
  auto gload = [&](int c0) {
    const bool qv = c0 + quad * 4 < p.c4;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) sreg[it] = *reinterpret_cast<const f4*>((((smask >> it) & 1u) && qv) ? p.x + soff[it] + c0 : g_zero);
  };
  auto sstore = [&](int buf, int c0) {
    float* tile = tile0 + buf * tile_floats;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = tid + it * 256;
      if (idx < slots) *reinterpret_cast<f4*>(tile + (idx >> 2) * PITCH + quad * 4) = sreg[it];
    }
    const int nq = min(4, (p.c4 - c0) >> 2);
    if (tid < TABN) {
      const int tap = tid / nq, qd = tid - tap * nq;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ks = k == 0 ? 5 : (k == 1 ? 3 : 1), taps = ks * ks;
        int off = 0;
        if (tid < ((taps * nq + 3) >> 2) * 4 && tap < taps) {
          const int ky = tap / ks, kx = tap - ky * ks, d = p.hl - (ks >> 1);
          off = ((d + ky) * p.tc + d + kx) * PITCH + qd * 4;
        }
        tab0[(buf * 3 + k) * TABN + tid] = off;
      }
    }
  };
  f4 accA[MT][NA > 0 ? NA : 1], accB[MT][NB > 0 ? NB : 1], accC[MT][NC > 0 ? NC : 1];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < (NA > 0 ? NA : 1); ++j) accA[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NB > 0 ? NB : 1); ++j) accB[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NC > 0 ? NC : 1); ++j) accC[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  }
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) abase[i] = ((2 * wave + i) * p.tc + lr) * PITCH;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[0]), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[1]), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[2]), 0, 0x7fffffff, 0x00020000);
  unsigned jbA[NA > 0 ? NA : 1], jbB[NB > 0 ? NB : 1], jbC[NC > 0 ? NC : 1];
#pragma unroll
  for (int j = 0; j < (NA > 0 ? NA : 1); ++j) jbA[j] = (unsigned)(min(j, max(p.nt_total[0], 1) - 1) * 64 + lane) * 16u;
#pragma unroll
  for (int j = 0; j < (NB > 0 ? NB : 1); ++j) jbB[j] = (unsigned)(min(j, max(p.nt_total[1], 1) - 1) * 64 + lane) * 16u;
#pragma unroll
  for (int j = 0; j < (NC > 0 ? NC : 1); ++j) jbC[j] = (unsigned)(min(j, max(p.nt_total[2], 1) - 1) * 64 + lane) * 16u;
  unsigned soA = 0, soB = 0, soC = 0;     // byte offsets of the current chunk in the three streams
  gload(0);
  sstore(0, 0);
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < p.c4; c0 += 16) {
    const bool more = c0 + 16 < p.c4;
    if (more) gload(c0 + 16);
    const int nq = min(4, (p.c4 - c0) >> 2);
    const float* tile = tile0 + buf * tile_floats;
    const int* tab = tab0 + buf * 3 * TABN + lq;
    if constexpr (NA > 0) {
      const int ngr = (25 * nq + 3) >> 2;
      mma_groups<MT, NA>(accA, tile, tab, abase, rA, jbA, soA, (unsigned)p.nt_total[0] * 1024u, ngr);
      soA += (unsigned)ngr * p.nt_total[0] * 1024u;
    }
    if constexpr (NB > 0) {
      const int ngr = (9 * nq + 3) >> 2;
      mma_groups<MT, NB>(accB, tile, tab + TABN, abase, rB, jbB, soB, (unsigned)p.nt_total[1] * 1024u, ngr);
      soB += (unsigned)ngr * p.nt_total[1] * 1024u;
    }
    if constexpr (NC > 0) {
      mma_groups<MT, NC>(accC, tile, tab + 2 * TABN, abase, rC, jbC, soC, (unsigned)p.nt_total[2] * 1024u, 1);
      soC += (unsigned)p.nt_total[2] * 1024u;
    }
    if (more) {
      sstore(buf ^ 1, c0 + 16);
      __syncthreads();
      buf ^= 1;
    }
  }
  __syncthreads();
  if constexpr (NA > 0) s1_epilogue<NA, STATS>(accA, p, 0, n, tt, oy0, ox0, wave, lr, lq, red);
  if constexpr (NB > 0) s1_epilogue<NB, STATS>(accB, p, 1, n, tt, oy0, ox0, wave, lr, lq, red);
  if constexpr (NC > 0) s1_epilogue<NC, STATS>(accC, p, 2, n, tt, oy0, ox0, wave, lr, lq, red);
}

// ------------------------------------------------------------------------------------------------ frozen teacher: stage 1 in one launch
// The first convs of an EVAL-mode InvertedResidualChannels block with BatchNorm folded (cat_amd/frozen.py; inception_modules.py:135-165): the
// 5 x 5 and 3 x 3 convs of the residual branches (256 -> 42 each) and the N-concatenated 1 x 1 convs of the k = 1 residual branch and the three
// depthwise branches (256 -> 176) from ONE staging of the input tile per 16-channel chunk, epilogue = bias + activation, three output
// buffers.  Round 6: they were three launches -- two LDS-tile convs staging the same 256-channel input and a 128 x 128-tile GEMM that is
// overhead-bound at K = 256 (8 chunks per workgroup, 31 % tile padding at N = 176).  Accumulators: (N5 + N3 + N1) x 2 tiles of 16 x 16 per wave;
// the wide 1 x 1 slot is walked in passes of <= 3 N tiles so that two operand sets stay 24 registers.
struct S1WArgs {
  const float* x; const float* pack[3]; const float* bias[3]; float* y[3];
  int ycs[3], nvalid[3];
  int xcs, c4, N, H, W, reflect, act;
  float slope;
  int hl, tr, tc, tiles_x, tiles;
};

template <int NT>
__device__ __forceinline__ void s1w_store(const f4 (&acc)[2][NT], const S1WArgs& p, int k, int t0, int n, int oy0, int ox0, int wave, int lr, int lq) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = oy0 + 2 * wave + i;
    if (oy >= p.H) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ox = ox0 + lq * 4 + rg;
      if (ox >= p.W) continue;
      float* yo = p.y[k] + (((int64_t)n * p.H + oy) * p.W + ox) * p.ycs[k];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = (t0 + j) * 16 + lr;
        if (co < p.nvalid[k]) yo[co] = cat::apply_act(acc[i][j][rg] + (p.bias[k] ? p.bias[k][co] : 0.f), p.act, p.slope);
        else if (co < p.ycs[k]) yo[co] = 0.f;
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void wload(f4 (&b)[3], const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&jb)[3], unsigned so) {
#pragma unroll
  for (int j = 0; j < NT; ++j) b[j] = bload(rsrc, jb[j], so);
}

template <int MT, int NT>
__device__ __forceinline__ void wmma(f4 (&acc)[MT][3], const f4 (&a)[MT], const f4 (&b)[3]) {
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][tq], b[j][tq], acc[i][j], 0, 0, 0);
}

template <int N5, int N3, int N1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tstage1w_kernel(const S1WArgs p) {
  constexpr int MT = 2, TW = 16, MAXIT = ((TH + 4) * (TW + 4) * 4 + 255) / 256;
  constexpr int P1 = (N1 + 2) / 3, R1 = N1 - 3 * (P1 - 1);      // passes over the 1 x 1 slot, tiles of the last one (1..3)
  static_assert(N1 >= 1 && N1 <= 12 && N5 >= 1 && N5 <= 3 && N3 >= 1 && N3 <= 3, "tile counts");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_floats = p.tr * p.tc * PITCH;
  float* tile0 = smem;
  int* tab0 = reinterpret_cast<int*>(smem + 2 * tile_floats);      // [buf][3][TABN]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int tt = cat::xcd_remap(blockIdx.x, gridDim.x);
  const int n = tt / p.tiles, t = tt - n * p.tiles;
  const int oy0 = (t / p.tiles_x) * TH, ox0 = (t % p.tiles_x) * TW;
  const int slots = p.tr * p.tc * 4, quad = tid & 3;
  unsigned soff[MAXIT], smask = 0;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int pix = (tid + it * 256) >> 2;
    const int r = pix / p.tc;
    int iy = oy0 - p.hl + r, ix = ox0 - p.hl + (pix - r * p.tc);
    bool v = tid + it * 256 < slots;
    if (p.reflect) {
      v = v && iy > -p.H && iy < 2 * p.H - 1 && ix > -p.W && ix < 2 * p.W - 1;
      iy = cat::reflect_idx(iy, p.H);
      ix = cat::reflect_idx(ix, p.W);
    } else {
      v = v && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    }
    smask |= v ? (1u << it) : 0u;
    soff[it] = v ? ((unsigned)(n * p.H + iy) * (unsigned)p.W + (unsigned)ix) * (unsigned)p.xcs + quad * 4 : 0u;
  }
  f4 sreg[MAXIT];
  auto gload = [&](int c0) {
    const bool qv = c0 + quad * 4 < p.c4;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) sreg[it] = *reinterpret_cast<const f4*>((((smask >> it) & 1u) && qv) ? p.x + soff[it] + c0 : g_zero);
  };
  auto sstore = [&](int buf, int c0) {
    float* tile = tile0 + buf * tile_floats;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = tid + it * 256;
      if (idx < slots) *reinterpret_cast<f4*>(tile + (idx >> 2) * PITCH + quad * 4) = sreg[it];
    }
    const int nq = min(4, (p.c4 - c0) >> 2);
    if (tid < TABN) {
      const int tap = tid / nq, qd = tid - tap * nq;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ks = k == 0 ? 5 : (k == 1 ? 3 : 1), taps = ks * ks;
        int off = 0;
        if (tid < ((taps * nq + 3) >> 2) * 4 && tap < taps) {
          const int ky = tap / ks, kx = tap - ky * ks, d = p.hl - (ks >> 1);
          off = ((d + ky) * p.tc + d + kx) * PITCH + qd * 4;
        }
        tab0[(buf * 3 + k) * TABN + tid] = off;
      }
    }
  };
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f4 acc5[MT][N5], acc3[MT][N3], accA[MT][3], accB[MT][3], accC[MT][3], accD[MT][3];      // the 1 x 1 slot: passes A..D of up to 3 tiles
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < N5; ++j) acc5[i][j] = zero4;
#pragma unroll
    for (int j = 0; j < N3; ++j) acc3[i][j] = zero4;
#pragma unroll
    for (int j = 0; j < 3; ++j) accA[i][j] = accB[i][j] = accC[i][j] = accD[i][j] = zero4;
  }
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) abase[i] = ((2 * wave + i) * p.tc + lr) * PITCH;
  const __amdgpu_buffer_rsrc_t r5 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[0]), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[1]), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pack[2]), 0, 0x7fffffff, 0x00020000);
  unsigned jb5[N5], jb3[N3], jb1[4][3];
#pragma unroll
  for (int j = 0; j < N5; ++j) jb5[j] = (unsigned)(j * 64 + lane) * 16u;
#pragma unroll
  for (int j = 0; j < N3; ++j) jb3[j] = (unsigned)(j * 64 + lane) * 16u;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 3; ++j) jb1[q][j] = (unsigned)(min(3 * q + j, N1 - 1) * 64 + lane) * 16u;
  unsigned so5 = 0, so3 = 0, so1 = 0;     // byte offsets of the current chunk in the three streams
  gload(0);
  sstore(0, 0);
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < p.c4; c0 += 16) {
    const bool more = c0 + 16 < p.c4;
    if (more) gload(c0 + 16);
    const int nq = min(4, (p.c4 - c0) >> 2);
    const float* tile = tile0 + buf * tile_floats;
    const int* tab = tab0 + buf * 3 * TABN + lq;
    f4 bw[3];
    wload<(P1 == 1 ? R1 : 3)>(bw, r1, jb1[0], so1);
    {
      const int ngr = (25 * nq + 3) >> 2;
      mma_groups<MT, N5>(acc5, tile, tab, abase, r5, jb5, so5, (unsigned)N5 * 1024u, ngr);
      so5 += (unsigned)ngr * N5 * 1024u;
    }
    {
      const int ngr = (9 * nq + 3) >> 2;
      mma_groups<MT, N3>(acc3, tile, tab + TABN, abase, r3, jb3, so3, (unsigned)N3 * 1024u, ngr);
      so3 += (unsigned)ngr * N3 * 1024u;
    }
    {
      // the 1 x 1 slot: ONE group per chunk (the centre tap's four quads), N1 tiles wide -- the A fragments are read once, the filter
      // blocks of pass q + 1 are in flight behind the MFMAs of pass q, those of pass 0 were requested at the top of the chunk
      f4 aw[MT], bx[3];
      const int off = tab[2 * TABN];
#pragma unroll
      for (int i = 0; i < MT; ++i) aw[i] = *reinterpret_cast<const f4*>(tile + abase[i] + off);
      if constexpr (P1 > 1) wload<(P1 == 2 ? R1 : 3)>(bx, r1, jb1[1], so1);
      __builtin_amdgcn_sched_barrier(0);
      wmma<MT, (P1 == 1 ? R1 : 3)>(accA, aw, bw);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (P1 > 1) {
        if constexpr (P1 > 2) wload<(P1 == 3 ? R1 : 3)>(bw, r1, jb1[2], so1);
        __builtin_amdgcn_sched_barrier(0);
        wmma<MT, (P1 == 2 ? R1 : 3)>(accB, aw, bx);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (P1 > 2) {
        if constexpr (P1 > 3) wload<R1>(bx, r1, jb1[3], so1);
        __builtin_amdgcn_sched_barrier(0);
        wmma<MT, (P1 == 3 ? R1 : 3)>(accC, aw, bw);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (P1 > 3) wmma<MT, R1>(accD, aw, bx);
    }
    so1 += (unsigned)N1 * 1024u;
    if (more) {
      sstore(buf ^ 1, c0 + 16);
      __syncthreads();
      buf ^= 1;
    }
  }
  s1w_store<N5>(acc5, p, 0, 0, n, oy0, ox0, wave, lr, lq);
  s1w_store<N3>(acc3, p, 1, 0, n, oy0, ox0, wave, lr, lq);
  s1w_store<3>(accA, p, 2, 0, n, oy0, ox0, wave, lr, lq);
  if constexpr (P1 > 1) s1w_store<3>(accB, p, 2, 3, n, oy0, ox0, wave, lr, lq);
  if constexpr (P1 > 2) s1w_store<3>(accC, p, 2, 6, n, oy0, ox0, wave, lr, lq);
  if constexpr (P1 > 3) s1w_store<3>(accD, p, 2, 9, n, oy0, ox0, wave, lr, lq);
}

// dst[(G * nt_total + j) * 256 + lane * 4 + e]: G enumerates the MFMA groups of all chunks of one segment in consumption order
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int mode, int Nn, int Ck, int ks,
                                                   int wcs, int wn, int c4, int nt_total, int64_t total4) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int lane = (int)(e & 63);
  const int64_t gj = e >> 6;
  const int j = (int)(gj % nt_total);
  int G = (int)(gj / nt_total);
  const int taps = ks * ks;
  const int nfull = c4 >> 4;
  int chunk, gi, nq;
  if (G < nfull * taps) {
    chunk = G / taps;
    gi = G - chunk * taps;
    nq = 4;
  } else {
    chunk = nfull;
    gi = G - nfull * taps;
    nq = (c4 & 15) >> 2;
  }
  const int lr = lane & 15, lq = lane >> 4;
  const int p = 4 * gi + lq;
  const int tap = p / nq, qd = p - tap * nq;
  const int nn = j * 16 + lr;
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (tap < taps && nn < Nn) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = chunk * 16 + qd * 4 + t;
      if (c < Ck) v[t] = mode == 0 ? w[(int64_t)nn * wn + tap * wcs + c] : w[(int64_t)c * wn + (taps - 1 - tap) * wcs + nn];
    }
  }
  *reinterpret_cast<f4*>(dst + e * 4) = v;
}

static int groups_of(int ks, int c4) {
  const int taps = ks * ks;
  const int nfull = c4 >> 4, rem = (c4 & 15) >> 2;
  return nfull * taps + (rem ? (taps * rem + 3) / 4 : 0);
}

}  // namespace cat_pk

extern "C" {

size_t cat_tconv_pack_floats(int ks, int c4, int Nn) {
  if (ks <= 0 || c4 <= 0 || (c4 & 3) || Nn <= 0) return 0;
  return (size_t)cat_pk::groups_of(ks, c4) * cat::cdiv(Nn, 16) * 256;
}

int cat_tconv_pack(const float* w, int mode, int Nn, int Ck, int ks, int wcs, int wn, int c4, float* dst, cat_stream_t stream) {
  CAT_REQUIRE(ks == 1 || ks == 3 || ks == 5, "tconv pack: kernel size %d unsupported", ks);
  CAT_REQUIRE(c4 > 0 && (c4 & 3) == 0 && Ck <= c4 && Nn > 0, "tconv pack: bad channel counts (c4=%d Ck=%d Nn=%d)", c4, Ck, Nn);
  CAT_REQUIRE(mode == 0 ? Ck <= wcs : Nn <= wcs, "tconv pack: filter rows shorter than the channels read");
  const int nt = cat::cdiv(Nn, 16);
  const int64_t total4 = (int64_t)cat_pk::groups_of(ks, c4) * nt * 64;
  cat_pk::pack_kernel<<<(int)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, dst, mode, Nn, Ck, ks, wcs, wn, c4, nt, total4);
  return cat::check_launch("tconv_pack");
}

int cat_tconv_fwd(const cat_tconv_t* g, const float* pack, const float* bias, float* y, cat_stream_t stream) {
  CAT_REQUIRE(g->nseg >= 1 && g->nseg <= CAT_TCONV_MAXSEG, "tconv: %d segments", g->nseg);
  CAT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0 && g->Nn > 0, "tconv: empty geometry");
  CAT_REQUIRE((g->ycs & 3) == 0 && g->ycs >= g->Nn && g->ycw <= g->ycs, "tconv: bad output stride");
  CAT_REQUIRE(g->res == nullptr || g->rcs >= g->Nn, "tconv: residual stride");
  int hl = 0, hr = 0;
  double kflops = 0.0;
  for (int s = 0; s < g->nseg; ++s) {
    const cat_tseg_t& sg = g->seg[s];
    CAT_REQUIRE(sg.ks == 1 || sg.ks == 3 || sg.ks == 5, "tconv: kernel size %d unsupported", sg.ks);
    CAT_REQUIRE(sg.c4 > 0 && (sg.c4 & 3) == 0 && (sg.xcs & 3) == 0 && sg.xcs >= sg.c4, "tconv: segment %d channel layout", s);
    CAT_REQUIRE(sg.padv >= 0 && sg.padv <= 4, "tconv: segment %d padv", s);
    CAT_REQUIRE(sg.act == CAT_ACT_NONE || sg.act == CAT_ACT_RELU || sg.act == CAT_ACT_LRELU, "tconv: staging activation %d", sg.act);
    CAT_REQUIRE(!sg.reflect || (sg.ks <= g->H && sg.ks <= g->W), "tconv: reflect padding wider than the plane");
    CAT_REQUIRE((int64_t)g->N * g->H * g->W * sg.xcs < (int64_t)4294967295LL, "tconv: source larger than 2^32 elements");
    hl = sg.padv > hl ? sg.padv : hl;
    hr = sg.ks - 1 - sg.padv > hr ? sg.ks - 1 - sg.padv : hr;   // stays >= 0: a segment whose taps all lie left of the origin needs none
    CAT_REQUIRE(sg.cin > 0 && sg.cin <= sg.c4, "tconv: segment %d valid channel count", s);
    kflops += (double)sg.ks * sg.ks * sg.cin;
  }
  CAT_REQUIRE(hl + hr <= 4, "tconv: halo %d + %d exceeds the staged patch", hl, hr);
  static const int tw_env = getenv("CAT_PK_TW") ? atoi(getenv("CAT_PK_TW")) : 0;
  cat_pk::Launch L;
  L.nt_total = cat::cdiv(g->Nn, 16);
  L.nblk = cat::cdiv(L.nt_total, 8);
  const int nt = cat::cdiv(L.nt_total, L.nblk);
  L.nblk = cat::cdiv(L.nt_total, nt);
  // 8 x 16 pixel tiles; 8 x 32 (half the halo traffic, twice the filter reuse per wave) only for a very large grid of ONE N tile: with
  // more N tiles the 4 x NT accumulator tiles cost occupancy (tconv_kernel<4, 32>: 231 registers, 2 waves / SIMD; <5..8, 32>: 257..337,
  // 1 wave / SIMD) -- measured on the GauGAN step (round 3): 8 x 32 for every NT 62.8-63.0, NT <= 2 64.3, NT == 1 64.0-66.9, never 63.6 images/s
  const int64_t wg16 = (int64_t)g->N * cat::cdiv(g->Ho, 8) * cat::cdiv(g->Wo, 16) * L.nblk;
  static const int tw32_maxnt = getenv("CAT_PK_TW32_MAXNT") ? atoi(getenv("CAT_PK_TW32_MAXNT")) : 1;
  static const int tw32_minwg = getenv("CAT_PK_TW32_MINWG") ? atoi(getenv("CAT_PK_TW32_MINWG")) : 4096;
  const int tw = g->stats ? 16 : (tw_env ? tw_env : (wg16 >= tw32_minwg && nt <= tw32_maxnt ? 32 : 16));
  CAT_REQUIRE(tw == 16 || tw == 32, "tconv: CAT_PK_TW must be 16 or 32");
  CAT_REQUIRE(g->stats == nullptr || (g->act == CAT_ACT_NONE && g->res == nullptr && g->scs >= g->ycw), "tconv: statistics need a plain epilogue");
  // diagnostic build only (tools/debug/tconv_ablate.py): read ONCE per process, and loud -- a non-zero value makes the results wrong by design
  static const int ablate_env = [] {
    const int v = (cat::kDiag && getenv("CAT_PK_ABLATE")) ? atoi(getenv("CAT_PK_ABLATE")) : 0;
    if (v) fprintf(stderr, "libcat_hip: CAT_PK_ABLATE=%d -- tconv results are INTENTIONALLY WRONG (timing diagnostics only)\n", v);
    return v;
  }();
  L.ablate = ablate_env;
  L.hl = hl;
  L.tr = cat_pk::TH + hl + hr;
  L.tc = tw + hl + hr;
  L.tiles_x = cat::cdiv(g->Wo, tw);
  L.tiles = L.tiles_x * cat::cdiv(g->Ho, cat_pk::TH);
  const int64_t grid = (int64_t)g->N * L.tiles * L.nblk;
  CAT_REQUIRE(grid < (int64_t)2147483647, "tconv: grid too large");
  const size_t lds = (size_t)2 * L.tr * L.tc * cat_pk::PITCH * sizeof(float) + 2 * cat_pk::TABN * sizeof(int) +
                     (g->stats ? (size_t)8 * nt * 16 * sizeof(float) : 0);
  CAT_REQUIRE(lds <= 96 * 1024, "tconv: %zu bytes of LDS (max 96 KB)", lds);
  hipStream_t s = (hipStream_t)stream;
  cat::ProfScope prof(g->nseg > 1 ? "conv_tconv_multi" : "conv_tconv", 2.0 * (double)g->N * g->Ho * g->Wo * (g->nvalid > 0 ? g->nvalid : g->Nn) * kflops, 0.0, stream);
#define CAT_PK_LAUNCH(NT, TW)                                                                                              \
  {                                                                                                                        \
    static cat::LdsOptIn optin;                                                                                            \
    cat::lds_optin(optin, (const void*)cat_pk::tconv_kernel<NT, TW>, 96 * 1024);                                           \
    cat_pk::tconv_kernel<NT, TW><<<(int)grid, 256, lds, s>>>(*g, pack, bias, y, L);                                        \
  }
#define CAT_PK_NT(TW)                          \
  switch (nt) {                                \
    case 1: CAT_PK_LAUNCH(1, TW) break;        \
    case 2: CAT_PK_LAUNCH(2, TW) break;        \
    case 3: CAT_PK_LAUNCH(3, TW) break;        \
    case 4: CAT_PK_LAUNCH(4, TW) break;        \
    case 5: CAT_PK_LAUNCH(5, TW) break;        \
    case 6: CAT_PK_LAUNCH(6, TW) break;        \
    case 7: CAT_PK_LAUNCH(7, TW) break;        \
    default: CAT_PK_LAUNCH(8, TW) break;       \
  }
  if (tw == 16) { CAT_PK_NT(16) } else { CAT_PK_NT(32) }
#undef CAT_PK_NT
#undef CAT_PK_LAUNCH
  return cat::check_launch("tconv_fwd");
}

static int tstage1_launch(const cat_tstage1_t* g, const float* x, const float* const* packs, const float* bias, float* const* ys,
                          const int* ycs3, float* stats, const char* what, cat_stream_t stream) {
  CAT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->cin > 0 && (g->xcs & 3) == 0 && g->xcs >= g->cin, "tstage1: bad input geometry");
  CAT_REQUIRE((int64_t)g->N * g->H * g->W * g->xcs < (int64_t)4294967295LL, "tstage1: source larger than 2^32 elements");
  cat_pk::S1Args a{};
  a.x = x; a.bias = bias; a.stats = stats;
  a.xcs = g->xcs; a.c4 = (g->cin + 3) & ~3; a.N = g->N; a.H = g->H; a.W = g->W; a.reflect = g->reflect; a.scs = g->scs;
  int nt[3];
  double kflops = 0.0;
  for (int k = 0; k < 3; ++k) {
    nt[k] = g->width[k] > 0 ? cat::cdiv(g->width[k], 16) : 0;
    a.pack[k] = nt[k] ? packs[k] : x;     // a valid address for the (unused) buffer resource
    a.ys[k] = ys[k]; a.ycs3[k] = ycs3[k];
    a.nt_total[k] = nt[k]; a.col0[k] = g->col0[k]; a.width[k] = g->width[k]; a.nvalid[k] = g->nvalid[k];
    CAT_REQUIRE(nt[k] == 0 || (packs[k] && ys[k] && (ycs3[k] & 3) == 0 && g->col0[k] + g->width[k] <= ycs3[k] && g->nvalid[k] <= g->width[k]),
                "tstage1: slice %d", k);
    const int ks = k == 0 ? 5 : (k == 1 ? 3 : 1);
    kflops += (double)g->nvalid[k] * ks * ks;
  }
  a.hl = nt[0] ? 2 : (nt[1] ? 1 : 0);
  CAT_REQUIRE(!g->reflect || (2 * a.hl + 1 <= g->H && 2 * a.hl + 1 <= g->W), "tstage1: reflect padding wider than the plane");
  a.tr = cat_pk::TH + 2 * a.hl;
  a.tc = 16 + 2 * a.hl;
  a.tiles_x = cat::cdiv(g->W, 16);
  a.tiles = a.tiles_x * cat::cdiv(g->H, cat_pk::TH);
  const int64_t grid = (int64_t)g->N * a.tiles;
  const int nmax = nt[0] > nt[1] ? (nt[0] > nt[2] ? nt[0] : nt[2]) : (nt[1] > nt[2] ? nt[1] : nt[2]);
  const size_t lds = (size_t)2 * a.tr * a.tc * cat_pk::PITCH * sizeof(float) + 6 * cat_pk::TABN * sizeof(int) + (size_t)8 * nmax * 16 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  cat::ProfScope prof(what, 2.0 * (double)g->N * g->H * g->W * g->cin * kflops, 0.0, stream);
#define CAT_S1(NA, NB, NC)                                                                        \
  if (nt[0] == NA && nt[1] == NB && nt[2] == NC) {                                                \
    if (stats) cat_pk::tstage1_kernel<NA, NB, NC, true><<<(int)grid, 256, lds, s>>>(a);           \
    else cat_pk::tstage1_kernel<NA, NB, NC, false><<<(int)grid, 256, lds, s>>>(a);                \
    return cat::check_launch(what);                                                               \
  }
  CAT_S1(1, 1, 1) CAT_S1(2, 1, 1) CAT_S1(1, 2, 1) CAT_S1(2, 2, 1)
  CAT_S1(1, 1, 2) CAT_S1(1, 1, 3) CAT_S1(1, 1, 4) CAT_S1(2, 1, 2) CAT_S1(2, 1, 3) CAT_S1(2, 1, 4)
  CAT_S1(1, 2, 2) CAT_S1(1, 2, 3) CAT_S1(1, 2, 4) CAT_S1(2, 2, 2) CAT_S1(2, 2, 3) CAT_S1(2, 2, 4)
#undef CAT_S1
  cat::set_error("tstage1: no kernel for %d / %d / %d N tiles (cat_tstage1_supported)", nt[0], nt[1], nt[2]);
  return -22;
}

int cat_tstage1_fwd(const cat_tstage1_t* g, const float* x, const float* const* packs, const float* bias, float* y, float* stats,
                    cat_stream_t stream) {
  CAT_REQUIRE(y && stats && (g->ycs & 3) == 0 && (g->scs & 3) == 0, "tstage1: output / statistics buffers");
  float* const ys[3] = {y, y, y};
  const int ycs3[3] = {g->ycs, g->ycs, g->ycs};
  return tstage1_launch(g, x, packs, bias, ys, ycs3, stats, "conv_tstage1", stream);
}

int cat_tstage1w_supported(int w5, int w3, int w1) {
  return w5 > 32 && w5 <= 48 && w3 > 32 && w3 <= 48 && w1 > 160 && w1 <= 176;      // the instantiation that exists: 3 + 3 + 11 tiles (m = 42)
}

int cat_tstage1w_fwd(const cat_tstage1w_t* g, const float* x, const float* const* packs, const float* const* biases, float* const* ys,
                     cat_stream_t stream) {
  CAT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->cin > 0 && (g->xcs & 3) == 0 && g->xcs >= g->cin, "tstage1w: bad input geometry");
  CAT_REQUIRE((int64_t)g->N * g->H * g->W * g->xcs < (int64_t)4294967295LL, "tstage1w: source larger than 2^32 elements");
  CAT_REQUIRE(cat_tstage1w_supported(g->nvalid[0], g->nvalid[1], g->nvalid[2]), "tstage1w: no kernel for %d / %d / %d output channels", g->nvalid[0],
              g->nvalid[1], g->nvalid[2]);
  CAT_REQUIRE(!g->reflect || (5 <= g->H && 5 <= g->W), "tstage1w: reflect padding wider than the plane");
  cat_pk::S1WArgs a{};
  a.x = x; a.xcs = g->xcs; a.c4 = (g->cin + 3) & ~3; a.N = g->N; a.H = g->H; a.W = g->W; a.reflect = g->reflect; a.act = g->act; a.slope = g->slope;
  double kflops = 0.0;
  for (int k = 0; k < 3; ++k) {
    CAT_REQUIRE(packs[k] && ys[k] && (g->ycs[k] & 3) == 0 && g->ycs[k] >= g->nvalid[k] && g->ycs[k] <= (k == 2 ? 176 : 48), "tstage1w: slot %d", k);
    a.pack[k] = packs[k]; a.bias[k] = biases ? biases[k] : nullptr; a.y[k] = ys[k]; a.ycs[k] = g->ycs[k]; a.nvalid[k] = g->nvalid[k];
    const int ks = k == 0 ? 5 : (k == 1 ? 3 : 1);
    kflops += (double)g->nvalid[k] * ks * ks;
  }
  a.hl = 2;
  a.tr = cat_pk::TH + 4;
  a.tc = 16 + 4;
  a.tiles_x = cat::cdiv(g->W, 16);
  a.tiles = a.tiles_x * cat::cdiv(g->H, cat_pk::TH);
  const int64_t grid = (int64_t)g->N * a.tiles;
  CAT_REQUIRE(grid < (int64_t)2147483647, "tstage1w: grid too large");
  const size_t lds = (size_t)2 * a.tr * a.tc * cat_pk::PITCH * sizeof(float) + 6 * cat_pk::TABN * sizeof(int);
  cat::ProfScope prof("conv_tstage1w", 2.0 * (double)g->N * g->H * g->W * g->cin * kflops, 0.0, stream);
  cat_pk::tstage1w_kernel<3, 3, 11><<<(int)grid, 256, lds, (hipStream_t)stream>>>(a);
  return cat::check_launch("tstage1w_fwd");
}

int cat_tstage1_supported(int w5, int w3, int w1) {
  const int a = cat::cdiv(w5, 16), b = cat::cdiv(w3, 16), c = cat::cdiv(w1, 16);
  return w5 > 0 && w3 > 0 && w1 > 0 && a <= 2 && b <= 2 && c >= 2 && c <= 4;
}

int cat_tstage1_dgrad(const cat_tstage1_t* g, const float* dy, const float* const* packs, float* const* dxs, const int* dxcs,
                      cat_stream_t stream) {
  CAT_REQUIRE(dxs && dxcs && !g->reflect, "tstage1 dgrad: zero-padded convolutions only, one output buffer per slot");
  return tstage1_launch(g, dy, packs, nullptr, dxs, dxcs, nullptr, "conv_tstage1_dgrad", stream);
}

int cat_tstage1_dgrad_supported(int w5, int w3, int w1) {
  const int a = cat::cdiv(w5, 16), b = cat::cdiv(w3, 16), c = cat::cdiv(w1, 16);
  return w5 > 0 && w3 > 0 && w1 > 0 && a <= 2 && b <= 2 && c <= 4;
}

}  // extern "C"
