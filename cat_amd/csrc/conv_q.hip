// Quad-granule LDS-tile convolution ("qconv") on v_mfma_f32_4x4x1_16B_f32, NHWC fp32, gfx950 (round 4).
//
//   out[n][cy*OS+py][cx*OS+px][co] = act( bias[co] + sum over K SEGMENTS s, taps (i, j) of s's kh x kw rectangle, channels c of s
//                                          f_s(src_s[n][cy*S + oy_s + i][cx*S + ox_s + j][c]) * W_s(tap, c, co) )
//
// Why another convolution kernel: the 16x16x4 MFMA of conv_pk.hip / conv_igemm.hip pads GEMM-N to 16 and GEMM-K to 4 per tap.  The layers
// of the pruned students are ragged in both (3 -> 22 7x7, 22 -> 37, 77 -> 17, 16 -> 3): 18 columns cost 32, the 3-channel image stem runs
// 196 x 32 MFMA work for 147 x 22 useful.  v_mfma_f32_4x4x1_16B computes sixteen independent 4 x 4 x 1 outer products per instruction at
// the SAME rate (tools/micro/mfma4x4.hip on the MI355X: 8.1 cycles per instruction per SIMD with two waves = 155 TFLOP/s), and its CBSZ / ABID
// operand broadcast turns it into a 4 (output channels) x 64 (pixels) x 1 (k) GEMM step whose A operand -- four filter values -- is taken
// from lanes 4*ABID .. 4*ABID+3 of ONE register: a register holds the filters of 16 k values for a quad of output channels, the N granule is
// 4 and the K granule is one (tap, channel quad) pair.
//
//   * lane <-> output pixel (a wave owns 4 rows x 16 columns of the tile), accumulator register i of quad q <-> output channel 4q + i;
//   * B operand = the lane's own source pixel: one ds_read_b128 from the staged patch gives 4 channels = 4 k steps for EVERY output quad
//     (4 * NQ MFMAs per LDS read, against 2 reads per 8 MFMAs in the 16x16x4 kernels);
//   * A operand = a pre-packed filter stream ([step][N split][quad group][lane][4], cat_qconv_pack): one buffer_load_dwordx4 per wave and
//     step feeds 4 output quads x 16 MFMAs;
//   * a STEP = 4 micro steps = 4 (tap, channel quad) pairs of the staged chunk.  Their LDS offsets come from a table that cat_qconv_pack
//     writes in front of the filters (one s_load_dwordx4 per step, requested two steps ahead): the first version computed them with ~50 scalar
//     instructions per step and was ISSUE-bound (a wave issues one instruction per 4 cycles: 16 MFMAs + 60 others per step; measured 0.3-0.5
//     of the MFMA rate, tools/debug/qconv_ablate.sh) -- now a step is 16 * NQ MFMAs + ~16 other instructions;
//   * the (tile + halo) source patch is staged per chunk of up to 48 channels (the whole K of the generator's edge layers in 1-2 chunks; double
//     buffered, one barrier per chunk), with the optional per-channel affine + activation of a preceding train-mode norm applied while staging;
//   * S = 2 (the stride-2 3x3 convs of the generator's down-sampling, inception_generator.py:44-46): the patch columns are stored parity-split
//     so that consecutive output pixels read consecutive LDS pixels for every tap (conflict-free ds_read_b128);
//   * ncls = 4 (ConvTranspose2d k3 s2 p1 op1, inception_generator.py:118-126): output pixels (2a+py, 2b+px) of sub-pixel class (py, px) are a
//     dense stride-1 correlation of the coarse input grid with the class's 1 / 2 / 2 / 4 taps -- one launch, class = a grid dimension;
//   * optional epilogue: bias, activation, residual, and per-tile (sum, sum of squared deviations from the tile mean) of the pre-activation
//     output for a following train-mode norm: the tile is transposed through LDS ([channel][pixel]) and every thread sums a strided slice of
//     one channel (the first version's 6-step shuffle reduction per value cost more than the layer's MFMAs); cat_tnorm_finalize2 merges the tiles.
#include "common.h"
#include <stdlib.h>

namespace cat_q {

constexpr int TW = 16;

struct Plan {
  int cs, pitch, nq, pg, ns, nblk, maxit;   // channels per staged chunk, floats per staged pixel, output quads per wave, pixel groups / N
                                            // splits per workgroup (pg * ns = 4), N blocks over the grid, staging iterations
  int hl, pr, pc, pch;                      // halo (top / left), staged rows / columns (even for S = 2), pc / 2
  int tiles_x, tiles;                       // tiles per row / per image of the output lattice
  int nsplit;                               // nblk * ns
  int nbuf;                                 // staged buffers (2 if any program has more than one chunk)
  int step0[4], steps_total;                // first step of each class's program in the stream; all steps
  int ftab;                                 // float offset of the filters inside the stream (behind the offset table)
  int nitems, grid;                         // work items = workgroups launched
  int ablate;                               // timing diagnostics (CAT_Q_ABLATE, results become WRONG): 1 every filter load re-reads the first
                                            // block (L1 hits), 4 no staging / barrier after the first chunk, 16 no output stores, 32 no statistics,
                                            // 64 one MFMA step per chunk, 128 no staging loads for the first chunk
};

struct KArgs {
  cat_qconv_t g;
  const float* pack;
  const float* bias;
  float* y;
  Plan p;
};

__host__ __device__ inline int nquads_of(int c4, int c0, int cs) { return ((c4 - c0 < cs ? c4 - c0 : cs)) >> 2; }
__host__ __device__ inline int steps_of_chunk(int ntaps, int nquads) { return (ntaps * nquads + 3) >> 2; }
__host__ __device__ inline int steps_of_seg(int ntaps, int c4, int cs) {
  int n = 0;
  for (int c0 = 0; c0 < c4; c0 += cs) n += steps_of_chunk(ntaps, nquads_of(c4, c0, cs));
  return n;
}

__device__ __attribute__((aligned(16))) float g_zero[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ f4 bload(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff_) {
  const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff_, 0);
  return __builtin_bit_cast(f4, r);
}

typedef int i4 __attribute__((ext_vector_type(4)));
// one table row (the LDS offsets of a step's 4 micro steps) through the scalar cache: uniform address, constant address space
__device__ __forceinline__ i4 tload(const int* tab, int step) {
  typedef __attribute__((address_space(4))) const i4 ci4;
  return *reinterpret_cast<ci4*>(reinterpret_cast<uintptr_t>(tab + 4 * (int64_t)step));
}

// tanh / ReLU6 epilogues (the image head): out of line, the expansion of tanhf is ~100 instructions per element
__device__ __attribute__((noinline)) f4 act4_slow(f4 v, int act, float slope) {
#pragma unroll 1
  for (int i = 0; i < 4; ++i) v[i] = cat::apply_act(v[i], act, slope);
  return v;
}

template <int ABID>
__device__ __forceinline__ f4 mm(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}

// the 4 * NQ MFMAs of micro step J: k slot 4J + c of the step's filter registers x channel c of the pixel vector.  NQ = 1 (the image head:
// 3 output channels) keeps one accumulator per channel-in-quad c, summed in the epilogue -- one accumulator would chain every MFMA to the
// previous one (the compiler separates dependent 4x4 MFMAs with wait states)
template <int J, int NQ, int NACC>
__device__ __forceinline__ void micro(f4 (&acc)[NACC], const f4 (&a)[(NQ + 3) / 4], const f4& b) {
  if constexpr (NQ == 1) {
    acc[0] = mm<4 * J + 0>(a[0][0], b[0], acc[0]);
    acc[1] = mm<4 * J + 1>(a[0][0], b[1], acc[1]);
    acc[2] = mm<4 * J + 2>(a[0][0], b[2], acc[2]);
    acc[3] = mm<4 * J + 3>(a[0][0], b[3], acc[3]);
    return;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = mm<4 * J + 0>(a[q >> 2][q & 3], b[0], acc[q]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = mm<4 * J + 1>(a[q >> 2][q & 3], b[1], acc[q]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = mm<4 * J + 2>(a[q >> 2][q & 3], b[2], acc[q]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = mm<4 * J + 3>(a[q >> 2][q & 3], b[3], acc[q]);
}

template <int NQ, int MAXIT>
__global__ __launch_bounds__(256) void qconv_kernel(const KArgs ka) {
  constexpr int NQ4 = (NQ + 3) / 4, NACC = NQ == 1 ? 4 : NQ;
  const cat_qconv_t& g = ka.g;
  const Plan& P = ka.p;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave % P.pg, ns = wave / P.pg;
  const int bid = cat::xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % P.nblk, tt = bid / P.nblk;              // tt = (image * tiles + tile) * ncls + class
  const int cls = g.ncls > 1 ? tt % g.ncls : 0;
  const int tl = g.ncls > 1 ? tt / g.ncls : tt;
  const int n = tl / P.tiles, t = tl - n * P.tiles;
  const int TH = 4 * P.pg;
  const int cy0 = (t / P.tiles_x) * TH, cx0 = (t % P.tiles_x) * TW;
  const int sfirst = g.ncls > 1 ? cls : 0, slast = g.ncls > 1 ? cls + 1 : g.nseg;
  const int PITCH = P.pitch, CS = P.cs, NCQ = CS >> 2;
  const int tile_floats = P.pr * P.pc * PITCH;
  const int npix = P.pr * P.pc;
  float* tile0 = smem;
  const int abl = cat::kDiag ? P.ablate : 0;      // diagnostic build only; compile-time 0 in the production kernels

  // staging map: slot = pixel * NCQ + quad; an iteration covers ACT = (256 / NCQ) * NCQ slots, so a thread keeps ONE channel quad (tid % NCQ)
  // through all its iterations -- the staging affine's scale / shift are two registers per chunk, loaded together with the data (the first
  // version re-loaded them per slot behind the data: +33 us on the 25 -> 40 stride-2 layer) -- and fewer than NCQ threads idle.
  // sdst: LDS float offset (-1: no slot); spq: source row + 64 (bits 0-13), source column + 64 (bits 14-27)
  int sdst[MAXIT];
  unsigned spq[MAXIT];
  const int dpix = 256 / NCQ, ACT = dpix * NCQ;
  const int quad = tid % NCQ;
  {
    const int drow = dpix / P.pc, dcol = dpix - drow * P.pc;
    int pix = tid / NCQ;
    int r = pix / P.pc, c = pix - r * P.pc;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int cslot = g.S == 2 ? (c & 1) * P.pch + (c >> 1) : c;
      const bool v = pix < npix && tid < ACT;
      sdst[it] = v ? (r * P.pc + cslot) * PITCH + quad * 4 : -1;
      spq[it] = (unsigned)(cy0 * g.S - P.hl + r + 64) | ((unsigned)(cx0 * g.S - P.hl + c + 64) << 14);
      pix += dpix;
      r += drow;
      c += dcol;
      if (c >= P.pc) {
        c -= P.pc;
        ++r;
      }
    }
  }
  unsigned soff[MAXIT];
  unsigned smask = 0;
  auto locate = [&](int s) {   // source offsets of the patch pixels for segment s
    const int refl = g.seg[s].reflect, xcs = g.seg[s].xcs;
    smask = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int iy = (int)(spq[it] & 0x3fffu) - 64, ix = (int)((spq[it] >> 14) & 0x3fffu) - 64;
      bool v = sdst[it] >= 0;
      if (refl) {
        v = v && iy > -g.H && iy < 2 * g.H - 1 && ix > -g.W && ix < 2 * g.W - 1;
        iy = cat::reflect_idx(iy, g.H);
        ix = cat::reflect_idx(ix, g.W);
      } else {
        v = v && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
      }
      smask |= v ? (1u << it) : 0u;
      soff[it] = v ? ((unsigned)(n * g.H + iy) * (unsigned)g.W + (unsigned)ix) * (unsigned)xcs + (unsigned)quad * 4u : 0u;   // < 2^32 elements (host-checked)
    }
  };
  f4 sreg[MAXIT], ssc = {1.f, 1.f, 1.f, 1.f}, ssh = {0.f, 0.f, 0.f, 0.f};
  bool s_qv = false;            // this thread's channel quad lies inside the segment's channels of the chunk
  int s_cur = 0;
  auto gload = [&](int s, int c0) {
    const cat_qseg_t& sg = g.seg[s];
    const float* src = sg.src;
    s_cur = s;
    s_qv = c0 + quad * 4 < sg.c4;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const bool v = ((smask >> it) & 1u) && s_qv;
      sreg[it] = *reinterpret_cast<const f4*>(v ? src + soff[it] + c0 : g_zero);
    }
    if (sg.scale) {
      const int so = n * sg.sstride + c0 + quad * 4;
      ssc = *reinterpret_cast<const f4*>(s_qv ? sg.scale + so : g_zero);
      ssh = *reinterpret_cast<const f4*>(s_qv ? sg.shift + so : g_zero);
    }
  };
  auto sstore = [&](int buf) {
    float* tile = tile0 + buf * tile_floats;
    const cat_qseg_t& sg = g.seg[s_cur];
    const bool aff = sg.scale != nullptr;
    const int act = sg.act;
    const float neg = act == CAT_ACT_RELU ? 0.f : (act == CAT_ACT_LRELU ? sg.slope : 1.f);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      f4 v = sreg[it];
      if (aff || act) {   // wave-uniform
        const bool ok = ((smask >> it) & 1u) && s_qv;   // padding pixels / channels stay exactly 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = aff ? fmaf(v[e], ssc[e], ssh[e]) : v[e];
          a = a > 0.f ? a : a * neg;      // neg: 0 (ReLU), slope (LeakyReLU), 1 (none)
          v[e] = ok ? a : 0.f;
        }
      }
      if (sdst[it] >= 0) *reinterpret_cast<f4*>(tile + sdst[it]) = v;
    }
  };

  const int ly = pg * 4 + (lane >> 4), lx = lane & 15;
  const int lbase = (ly * g.S * P.pc + lx) * PITCH;
  const int wsplit = nb * P.ns + ns;
  const int co0 = wsplit * NQ * 4;                 // this wave's first output channel
  // accumulators start at the bias (one scalar load per output quad; entries of a partial last quad beyond Nn read whatever follows the bias
  // vector -- those channels are masked at the store and in the statistics)
  f4 acc[NACC];
  {
    typedef __attribute__((address_space(4))) const f4 cf4;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      f4 bq = {0.f, 0.f, 0.f, 0.f};
      if (ka.bias && q < NQ && co0 + q * 4 < g.Nn) bq = *reinterpret_cast<cf4*>(reinterpret_cast<uintptr_t>(ka.bias + co0 + q * 4));
      acc[q] = bq;
    }
  }
  // filter stream of this wave: [step][N split][quad group][lane][4]; per-lane byte offset in voffset, the step's offset in soffset (scalar)
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ka.pack), 0, 0x7fffffff, 0x00020000);
  const int* tab = reinterpret_cast<const int*>(ka.pack);
  const unsigned avoff = (unsigned)((wsplit * NQ4 * 64 + lane) * 16);
  const unsigned astep = (unsigned)(P.nsplit * NQ4 * 1024);    // bytes per step
  auto aload = [&](f4 (&a)[NQ4], unsigned pos) {
#pragma unroll
    for (int v = 0; v < NQ4; ++v) a[v] = bload(prsrc, avoff + (unsigned)v * 1024u, pos);
  };
  auto bread = [&](f4 (&b)[4], const float* tile, const i4 o) {
    b[0] = *reinterpret_cast<const f4*>(tile + o[0]);
    b[1] = *reinterpret_cast<const f4*>(tile + o[1]);
    b[2] = *reinterpret_cast<const f4*>(tile + o[2]);
    b[3] = *reinterpret_cast<const f4*>(tile + o[3]);
  };

  // cursor over (segment, chunk); gstep = index of the current step in the stream (offset table row and filter block)
  int cs_ = sfirst, cc0 = 0;
  int gstep = P.step0[cls];
  unsigned apos = (unsigned)P.ftab * 4u + (unsigned)gstep * astep;       // byte offset of the current step's filters
  const unsigned apos0 = apos;
  f4 a[NQ4], an[NQ4];
#pragma unroll
  for (int v = 0; v < NQ4; ++v) a[v] = an[v] = bload(prsrc, avoff + (unsigned)v * 1024u, apos);
  i4 ocur = tload(tab, gstep);
  locate(sfirst);
  if (abl & 128) smask = 0;
  gload(sfirst, 0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  while (true) {
    const cat_qseg_t& sg = g.seg[cs_];
    const int nsteps = (abl & 64) ? 1 : steps_of_chunk(sg.kh * sg.kw, nquads_of(sg.c4, cc0, CS));
    // next chunk
    int nsn = cs_, nc0 = cc0 + CS;
    if (nc0 >= sg.c4) {
      ++nsn;
      nc0 = 0;
      if (nsn < slast) locate(nsn);
    }
    const bool more = nsn < slast;
    if (more && !(abl & 4)) gload(nsn, nc0);   // in flight behind this chunk's MFMA stream

    const float* tile = tile0 + buf * tile_floats + lbase;
    // two operand register sets in ping-pong (no copies inside the loop): the filters / pixel vectors of step st + 1 are requested BEFORE the
    // 16 * NQ MFMAs of step st issue, the offset row of step st + 2 too.  Beyond the chunk's last step the requests are harmless (the table
    // and the filter stream carry spare steps; the pixel vectors are re-read after the barrier).
    f4 b0[4], b1[4];
    bread(b0, tile, ocur);
    i4 onext = tload(tab, gstep + 1);
    int st = 0;
    while (true) {
      {
        aload(an, (abl & 1) ? apos0 : apos + astep);
        bread(b1, tile, onext);
        ocur = tload(tab, gstep + 2);
        __builtin_amdgcn_sched_barrier(0);
        micro<0, NQ, NACC>(acc, a, b0[0]);
        micro<1, NQ, NACC>(acc, a, b0[1]);
        micro<2, NQ, NACC>(acc, a, b0[2]);
        micro<3, NQ, NACC>(acc, a, b0[3]);
        __builtin_amdgcn_sched_barrier(0);
        apos += astep;
        ++gstep;
      }
      if (++st >= nsteps) {
#pragma unroll
        for (int v = 0; v < NQ4; ++v) a[v] = an[v];
        ocur = onext;
        break;
      }
      {
        aload(a, (abl & 1) ? apos0 : apos + astep);
        bread(b0, tile, ocur);
        onext = tload(tab, gstep + 2);
        __builtin_amdgcn_sched_barrier(0);
        micro<0, NQ, NACC>(acc, an, b1[0]);
        micro<1, NQ, NACC>(acc, an, b1[1]);
        micro<2, NQ, NACC>(acc, an, b1[2]);
        micro<3, NQ, NACC>(acc, an, b1[3]);
        __builtin_amdgcn_sched_barrier(0);
        apos += astep;
        ++gstep;
      }
      if (++st >= nsteps) break;
    }
    if (!more) break;
    if (!(abl & 4)) {
      sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    cs_ = nsn;
    cc0 = nc0;
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------------------------------------------
  if constexpr (NQ == 1) acc[0] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  const int cy = cy0 + ly, cx = cx0 + lx;
  const bool pv = cy < g.Ho && cx < g.Wo;
  if (g.stats && !(abl & 32)) {
    // per-tile sum and sum of squared deviations from the TILE mean of the pre-activation output.  The workgroup's tile goes to LDS as
    // [channel][pixel] (lanes = consecutive pixels: conflict-free); a channel is then summed by SEG threads, each over a stride-SEG slice
    // of float4s, in two passes (mean, squared deviations: no E[x^2] - E[x]^2 cancellation).
    __syncthreads();                               // the staged tiles are dead: reuse their LDS
    const int tp = 64 * P.pg, tpp = tp + 4;        // pixels per channel row (+ pad)
    const int wch = NQ * 4, tch = P.ns * wch;      // channels of a wave / of the workgroup
    float* tr = smem;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) tr[(ns * wch + q * 4 + i) * tpp + pg * 64 + lane] = acc[q][i];
    }
    __syncthreads();
    int seg = 1;
    while (seg * 2 * tch <= 256 && seg < 16) seg *= 2;          // threads per channel (power of two, <= 16)
    const int vrows = min(TH, g.Ho - cy0), vcols = min(TW, g.Wo - cx0);
    const int cnt = vrows * vcols;
    for (int ch = tid / seg; ch < tch; ch += 256 / seg) {
      const int sgi = tid & (seg - 1);
      const float* row = tr + ch * tpp;
      float s = 0.f;
      for (int p4 = sgi; p4 < tp / 4; p4 += seg) {
        const f4 v = *reinterpret_cast<const f4*>(row + p4 * 4);
        const bool rv = (p4 >> 2) < vrows;
        const int c0 = (p4 & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (rv && c0 + e < vcols) ? v[e] : 0.f;
      }
      for (int o = seg >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s / (float)cnt;
      float m2 = 0.f;
      for (int p4 = sgi; p4 < tp / 4; p4 += seg) {
        const f4 v = *reinterpret_cast<const f4*>(row + p4 * 4);
        const bool rv = (p4 >> 2) < vrows;
        const int c0 = (p4 & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - mean;
          m2 += (rv && c0 + e < vcols) ? d * d : 0.f;
        }
      }
      for (int o = seg >> 1; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
      if (sgi == 0) {
        // workgroup channel ch = (N split ns_, wave-local channel): its output channel
        const int ns_ = ch / wch, cl = ch - ns_ * wch;
        const int co = (nb * P.ns + ns_) * wch + cl;
        if (co < g.ycw) {
          float* dst = g.stats + (int64_t)tt * 2 * g.scs;
          const bool cv = co < g.Nn;
          dst[co] = cv ? s : 0.f;
          dst[g.scs + co] = cv ? m2 : 0.f;
        }
      }
    }
  }
  if (pv && !(abl & 16)) {
    const int py = cls >> 1, px = cls & 1;
    const int64_t pix = ((int64_t)n * (g.Ho * g.OS) + cy * g.OS + py) * (g.Wo * g.OS) + cx * g.OS + px;
    float* yo = ka.y + pix * g.ycs + co0;
    const float* ro = g.res ? g.res + pix * g.rcs + co0 : nullptr;
    const int act = g.act, Nn = g.Nn;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int co = co0 + q * 4;
      if (co >= g.ycw) break;      // uniform
      f4 v = acc[q];
      if (act == CAT_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      } else if (act == CAT_ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * g.slope;
      } else if (act != CAT_ACT_NONE) {
        v = act4_slow(v, act, g.slope);
      }
      if (co + 4 > Nn) {         // padding channels of the last quad are written as 0
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = co + i < Nn ? v[i] : 0.f;
      }
      if (ro) {
        const f4 rr = *reinterpret_cast<const f4*>(ro + q * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += co + i < Nn ? rr[i] : 0.f;
      }
      *reinterpret_cast<f4*>(yo + q * 4) = v;
    }
  }
}

// One segment's part of the stream.  Steps [sbase, sbase + steps_of_seg) of the program:
//   offset table  tab[step * 4 + m]            = LDS float offset of micro step m's pixel vector (relative to the lane's base)
//   filters       flt[((step * nsplit + wsplit) * NQ4 + q4) * 256 + lane * 4 + e]
// Micro step u of a chunk (c0, nquads) = (tap u / nquads, channel quad u % nquads); lane -> k slot lane >> 2 = (micro step, channel in quad),
// output channel ((wsplit * nq + 4 q4 + e) * 4 + (lane & 3)).  Source element of (output channel co, tap t, channel c):
// w[co * s_co + tapsrc[t] * s_tap + c * s_ci].
struct PackArgs {
  const float* w;
  float* dst;
  int Nn, Cin, c4, kh, kw, oy, ox, S;
  int cs, pitch, nq, nsplit, hl, pc, pch;
  int sbase, nsteps, ftab;
  int s_co, s_tap, s_ci;
  int tapsrc[49];
};

__global__ __launch_bounds__(256) void qpack_kernel(const PackArgs p) {
  const int nq4 = (p.nq + 3) / 4;
  const int ntaps = p.kh * p.kw;
  const int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nflt = (int64_t)p.nsteps * p.nsplit * nq4 * 64;
  // (step of the segment) -> (chunk, step in chunk)
  auto chunk_of = [&](int sl, int& c0, int& nquads, int& st) {
    c0 = 0;
    while (true) {
      nquads = nquads_of(p.c4, c0, p.cs);
      const int ns_ = steps_of_chunk(ntaps, nquads);
      if (sl < ns_) break;
      sl -= ns_;
      c0 += p.cs;
    }
    st = sl;
  };
  if (e4 < nflt) {
    const int lane = (int)(e4 & 63);
    int64_t r = e4 >> 6;
    const int q4 = (int)(r % nq4);
    r /= nq4;
    const int wsplit = (int)(r % p.nsplit);
    const int sl = (int)(r / p.nsplit);
    int c0, nquads, st;
    chunk_of(sl, c0, nquads, st);
    const int slot = lane >> 2, i = lane & 3;
    const int u = st * 4 + (slot >> 2);
    const int tap = u / nquads, c = c0 + (u - tap * nquads) * 4 + (slot & 3);
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (tap < ntaps && c < p.Cin) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q4 * 4 + e;
        const int co = (wsplit * p.nq + q) * 4 + i;
        if (q < p.nq && co < p.Nn) v[e] = p.w[(int64_t)co * p.s_co + (int64_t)p.tapsrc[tap] * p.s_tap + (int64_t)c * p.s_ci];
      }
    }
    *reinterpret_cast<f4*>(p.dst + p.ftab + ((int64_t)p.sbase * p.nsplit * nq4 * 64 + e4) * 4) = v;
  } else if (e4 < nflt + (int64_t)p.nsteps * 4) {
    const int k = (int)(e4 - nflt);
    const int sl = k >> 2, m = k & 3;
    int c0, nquads, st;
    chunk_of(sl, c0, nquads, st);
    const int u = st * 4 + m;
    int off = 0;
    if (u < ntaps * nquads) {
      const int tap = u / nquads, quad = u - tap * nquads;
      const int ti = tap / p.kw, tj = tap - ti * p.kw;
      const int col = p.hl + p.ox + tj;
      const int cslot = p.S == 2 ? (col & 1) * p.pch + (col >> 1) : col;
      off = ((p.hl + p.oy + ti) * p.pc + cslot) * p.pitch + quad * 4;
    }
    reinterpret_cast<int*>(p.dst)[(p.sbase + sl) * 4 + m] = off;
  }
}

static int g_min_tiles16 = 1024;      // cat_qconv_min_tiles16
static const int NQ_SET[] = {1, 2, 3, 4, 5, 6, 8, 10, 12};

static int make_plan(const cat_qconv_t* g, Plan* P) {
  if (!(g->nseg >= 1 && g->nseg <= CAT_QCONV_MAXSEG)) return -1;
  if (!(g->S == 1 || g->S == 2) || !(g->OS == 1 || g->OS == 2) || !(g->ncls == 1 || g->ncls == 4)) return -1;
  if (g->ncls == 4 && (g->nseg != 4 || g->OS != 2)) return -1;
  if (g->ncls == 1 && g->OS != 1) return -1;
  int c4max = 0, lo = 0, hi = 0;
  for (int s = 0; s < g->nseg; ++s) {
    const cat_qseg_t& sg = g->seg[s];
    if (sg.kh < 1 || sg.kw < 1 || sg.kh * sg.kw > 49 || sg.c4 <= 0 || (sg.c4 & 3)) return -1;
    c4max = sg.c4 > c4max ? sg.c4 : c4max;
    lo = -sg.oy > lo ? -sg.oy : lo;
    lo = -sg.ox > lo ? -sg.ox : lo;
    hi = sg.oy + sg.kh - 1 > hi ? sg.oy + sg.kh - 1 : hi;
    hi = sg.ox + sg.kw - 1 > hi ? sg.ox + sg.kw - 1 : hi;
  }
  const int Q = cat::cdiv(g->Nn, 4);
  const int64_t t16 = (int64_t)g->N * cat::cdiv(g->Ho, 16) * cat::cdiv(g->Wo, 16) * g->ncls;
  // 16 x 16 lattice tiles (4 pixel groups, every wave owns all output quads) for narrow outputs on planes large enough to fill the chip
  // with them; otherwise 8 x 16 tiles with the output quads split over two waves
  const int pg = (Q <= 8 && t16 >= g_min_tiles16 && g->S == 1) ? 4 : 2;
  P->pg = pg;
  P->ns = 4 / pg;
  int nblk = 1;
  while (cat::cdiv(Q, P->ns * nblk) > 12) ++nblk;      // <= 12 quads per wave: ~200 registers, two waves per SIMD
  const int nq = cat::cdiv(Q, P->ns * nblk);
  int nqt = 0;
  for (int k = 0; k < (int)(sizeof(NQ_SET) / sizeof(int)); ++k)
    if (NQ_SET[k] >= nq) {
      nqt = NQ_SET[k];
      break;
    }
  P->nq = nqt;
  P->nblk = nblk;
  P->nsplit = nblk * P->ns;
  P->hl = lo;
  const int th = 4 * pg;
  P->pr = (th - 1) * g->S + 1 + lo + hi;
  P->pc = (TW - 1) * g->S + 1 + lo + hi;
  if (g->S == 2) P->pc = (P->pc + 1) & ~1;
  P->pch = P->pc / 2;
  P->tiles_x = cat::cdiv(g->Wo, TW);
  P->tiles = P->tiles_x * cat::cdiv(g->Ho, th);
  // channels per staged chunk: as many as fit 2048 staging slots and ~40 KB per buffer (the whole K of an edge layer in one or two chunks:
  // every chunk costs a staging round trip and a barrier), split evenly; floats per staged pixel = cs rounded so that consecutive pixels fall
  // on different LDS bank quads (pitch / 4 odd) for the ds_read_b128 of 16 consecutive pixels
  const int npix = P->pr * P->pc;
  int cap = 48;
  while (cap > 4 && (npix * (cap / 4) > 8 * ((256 / (cap / 4)) * (cap / 4)) || (int64_t)npix * (cap + 4) * 4 > 40 * 1024)) cap -= 4;
  const int chunks = cat::cdiv(c4max, cap);
  P->cs = cat::round_up(cat::cdiv(c4max, chunks), 4);
  P->pitch = ((P->cs / 4) & 1) ? P->cs : P->cs + 4;
  if (P->cs == 4) P->pitch = 4;
  const int act_threads = (256 / (P->cs / 4)) * (P->cs / 4);      // threads that stage (a thread keeps one channel quad)
  const int it = cat::cdiv(npix * (P->cs / 4), act_threads);
  P->maxit = it <= 2 ? 2 : (it <= 4 ? 4 : 8);
  if (it > 8) return -2;
  // programs: one per class (ncls = 4: one segment each) or one over all segments
  int total = 0, multi = 0;
  for (int c = 0; c < 4; ++c) P->step0[c] = 0;
  if (g->ncls > 1) {
    for (int c = 0; c < g->ncls; ++c) {
      P->step0[c] = total;
      total += steps_of_seg(g->seg[c].kh * g->seg[c].kw, g->seg[c].c4, P->cs);
      multi |= g->seg[c].c4 > P->cs;
    }
  } else {
    int nch = 0;
    for (int s = 0; s < g->nseg; ++s) {
      total += steps_of_seg(g->seg[s].kh * g->seg[s].kw, g->seg[s].c4, P->cs);
      nch += cat::cdiv(g->seg[s].c4, P->cs);
    }
    multi = nch > 1;
  }
  P->steps_total = total;
  P->nitems = g->N * P->tiles * g->ncls * P->nblk;
  // one work item (image, tile, class, N block) per workgroup.  A PERSISTENT variant (a workgroup walks several items and requests item
  // k + 1's patch under item k's MFMA stream) was built and measured equal or slower on every layer (history: round 4): on gfx950 output
  // stores and operand loads share one in-order counter, so the next item's filter loads wait for the previous item's stores, and the
  // per-item address arithmetic is issue time of the same wave, not latency that overlaps
  P->grid = P->nitems;
  P->nbuf = multi ? 2 : 1;
  P->ftab = cat::round_up((total + 2) * 4, 64);      // two spare rows: the kernel requests offsets up to two steps ahead
  static const int ablate_env = [] {
    const int v = (cat::kDiag && getenv("CAT_Q_ABLATE")) ? atoi(getenv("CAT_Q_ABLATE")) : 0;
    if (v) fprintf(stderr, "libcat_hip: CAT_Q_ABLATE=%d -- qconv results are INTENTIONALLY WRONG (timing diagnostics only)\n", v);
    return v;
  }();
  P->ablate = ablate_env;
  return 0;
}

static int64_t stream_floats(const Plan& P) {
  return (int64_t)P.ftab + (int64_t)(P.steps_total + 1) * P.nsplit * ((P.nq + 3) / 4) * 256;      // one spare filter step
}

static size_t lds_bytes(const cat_qconv_t* g, const Plan& P) {
  const size_t tiles = (size_t)P.nbuf * P.pr * P.pc * P.pitch * sizeof(float);
  const size_t red = g->stats ? (size_t)P.ns * P.nq * 4 * (64 * P.pg + 4) * sizeof(float) : 0;
  return tiles > red ? tiles : red;
}

}  // namespace cat_q

extern "C" {

int cat_qconv_min_tiles16(int v) {
  const int old = cat_q::g_min_tiles16;
  if (v >= 0) cat_q::g_min_tiles16 = v;
  return old;
}

int cat_qconv_plan(const cat_qconv_t* g, cat_qplan_t* out) {
  cat_q::Plan P{};
  const int rc = cat_q::make_plan(g, &P);
  CAT_REQUIRE(rc == 0, "qconv plan: unsupported geometry (%d)", rc);
  out->cs = P.cs;
  out->nq = P.nq;
  out->nsplit = P.nsplit;
  out->th = 4 * P.pg;
  out->tw = cat_q::TW;
  out->tiles = P.tiles * g->ncls;
  out->pack_floats = cat_q::stream_floats(P);
  return 0;
}

int cat_qconv_pack(const cat_qconv_t* g, int seg, const float* w, float* dst, int Nn_w, const int* tapsrc, int s_co, int s_tap, int s_ci,
                   cat_stream_t stream) {
  cat_q::Plan P{};
  const int rc = cat_q::make_plan(g, &P);
  CAT_REQUIRE(rc == 0, "qconv pack: unsupported geometry (%d)", rc);
  CAT_REQUIRE(seg >= 0 && seg < g->nseg, "qconv pack: segment %d", seg);
  const cat_qseg_t& sg = g->seg[seg];
  CAT_REQUIRE(Nn_w > 0 && Nn_w <= g->Nn && g->Nn <= P.nsplit * P.nq * 4, "qconv pack: output channels");
  cat_q::PackArgs p{};
  p.w = w; p.dst = dst; p.Nn = Nn_w; p.Cin = sg.cin; p.c4 = sg.c4; p.kh = sg.kh; p.kw = sg.kw; p.oy = sg.oy; p.ox = sg.ox; p.S = g->S;
  p.cs = P.cs; p.pitch = P.pitch; p.nq = P.nq; p.nsplit = P.nsplit; p.hl = P.hl; p.pc = P.pc; p.pch = P.pch;
  p.s_co = s_co; p.s_tap = s_tap; p.s_ci = s_ci;
  const int ntaps = sg.kh * sg.kw;
  for (int t = 0; t < ntaps; ++t) p.tapsrc[t] = tapsrc ? tapsrc[t] : t;
  int sbase = 0;
  if (g->ncls > 1) sbase = P.step0[seg];
  else
    for (int s = 0; s < seg; ++s) sbase += cat_q::steps_of_seg(g->seg[s].kh * g->seg[s].kw, g->seg[s].c4, P.cs);
  p.sbase = sbase;
  p.nsteps = cat_q::steps_of_seg(ntaps, sg.c4, P.cs);
  p.ftab = P.ftab;
  const int64_t total = (int64_t)p.nsteps * P.nsplit * ((P.nq + 3) / 4) * 64 + (int64_t)p.nsteps * 4;
  cat_q::qpack_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(p);
  return cat::check_launch("qconv_pack");
}

int cat_qconv_fwd(const cat_qconv_t* g, const float* pack, const float* bias, float* y, cat_stream_t stream) {
  CAT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0 && g->Nn > 0, "qconv: empty geometry");
  CAT_REQUIRE(g->H < 8000 && g->W < 8000, "qconv: planes up to 8000 x 8000");
  CAT_REQUIRE((int64_t)g->N * g->Ho * g->OS * g->Wo * g->OS * g->ycs * 4 < (int64_t)2147483647, "qconv: output larger than 2 GB");
  CAT_REQUIRE((g->ycs & 3) == 0 && g->ycs >= g->Nn && g->ycw <= g->ycs && (g->ycw & 3) == 0, "qconv: bad output stride");
  CAT_REQUIRE(g->res == nullptr || (g->rcs >= g->ycw && (g->rcs & 3) == 0 && g->ncls == 1), "qconv: residual layout");
  CAT_REQUIRE(g->stats == nullptr || (g->act == CAT_ACT_NONE && g->res == nullptr && g->scs >= g->ycw), "qconv: statistics need a plain epilogue");
  cat_q::KArgs ka{};
  const int rc = cat_q::make_plan(g, &ka.p);
  CAT_REQUIRE(rc == 0, "qconv: unsupported geometry (%d)", rc);
  const cat_q::Plan& P = ka.p;
  double kflops = 0.0;
  for (int s = 0; s < g->nseg; ++s) {
    const cat_qseg_t& sg = g->seg[s];
    CAT_REQUIRE((sg.xcs & 3) == 0 && sg.xcs >= sg.c4 && sg.cin > 0 && sg.cin <= sg.c4, "qconv: segment %d channel layout", s);
    CAT_REQUIRE(sg.act == CAT_ACT_NONE || sg.act == CAT_ACT_RELU || sg.act == CAT_ACT_LRELU, "qconv: staging activation %d", sg.act);
    CAT_REQUIRE(!sg.reflect || (-sg.oy < g->H && -sg.ox < g->W && sg.oy + sg.kh - 1 < g->H && sg.ox + sg.kw - 1 < g->W),
                "qconv: reflect padding wider than the plane");
    CAT_REQUIRE((int64_t)g->N * g->H * g->W * sg.xcs < (int64_t)4294967295LL, "qconv: source larger than 2^32 elements");
    kflops += (double)sg.kh * sg.kw * sg.cin;
  }
  if (g->ncls > 1) kflops /= g->ncls;      // every class segment covers a quarter of the output pixels
  ka.g = *g;
  ka.pack = pack;
  ka.bias = bias;
  ka.y = y;
  CAT_REQUIRE((int64_t)g->N * P.tiles * g->ncls * P.nblk < (int64_t)2147483647, "qconv: too many work items");
  const int grid = P.grid;
  const size_t lds = cat_q::lds_bytes(g, P);
  CAT_REQUIRE(lds <= 96 * 1024, "qconv: %zu bytes of LDS (max 96 KB)", lds);
  hipStream_t s = (hipStream_t)stream;
  cat::ProfScope prof(g->nseg > 1 && g->ncls == 1 ? "conv_qconv_multi" : "conv_qconv",
                      2.0 * (double)g->N * g->Ho * g->Wo * g->ncls * (g->nvalid > 0 ? g->nvalid : g->Nn) * kflops, 0.0, stream);
#define CAT_Q_LAUNCH(NQ, MAXIT)                                                                    \
  {                                                                                                \
    static cat::LdsOptIn optin;                                                                    \
    cat::lds_optin(optin, (const void*)cat_q::qconv_kernel<NQ, MAXIT>, 96 * 1024);                 \
    cat_q::qconv_kernel<NQ, MAXIT><<<grid, 256, lds, s>>>(ka);                                \
  }
#define CAT_Q_NQ(MAXIT)                             \
  switch (P.nq) {                                   \
    case 1: CAT_Q_LAUNCH(1, MAXIT) break;           \
    case 2: CAT_Q_LAUNCH(2, MAXIT) break;           \
    case 3: CAT_Q_LAUNCH(3, MAXIT) break;           \
    case 4: CAT_Q_LAUNCH(4, MAXIT) break;           \
    case 5: CAT_Q_LAUNCH(5, MAXIT) break;           \
    case 6: CAT_Q_LAUNCH(6, MAXIT) break;           \
    case 8: CAT_Q_LAUNCH(8, MAXIT) break;           \
    case 10: CAT_Q_LAUNCH(10, MAXIT) break;         \
    default: CAT_Q_LAUNCH(12, MAXIT) break;         \
  }
  if (P.maxit == 2) { CAT_Q_NQ(2) }
  else if (P.maxit == 4) { CAT_Q_NQ(4) }
  else { CAT_Q_NQ(8) }
#undef CAT_Q_NQ
#undef CAT_Q_LAUNCH
  return cat::check_launch("qconv_fwd");
}

}  // extern "C"
