// K-concatenated SUM of stride-1 "same" convolutions as ONE implicit GEMM on 128 x 128 tiles, both operands DMA-ed into LDS, NHWC fp32, gfx950.
//
//   y[m][co] = act( bias[co] + sum over segments s, taps (ky, kx) of s, channels c of s
//                              src_s[pixel m shifted by (ky - pad_s, kx - pad_s)][c] * w_s[co][ky][kx][c] ) + res[m][co]
//
// This is the frozen teacher's block tail (cat_amd/frozen.py; models/modules/inception_modules.py:230-236 with every eval-mode BatchNorm
// folded): out = x + [h | hid3 | hid5] * [F ; W3 ; W5] + bias, i.e. a GEMM with M = N*H*W pixels, N = 256 output channels and
// K = 176 + 9*44 + 25*44 = 1672.  The LDS-tile kernel (conv_pk.hip) served it at 0.73 of the matrix pipe: its 1 x 1 segments are one
// 64-MFMA group per barrier, and its filter fragments come through VGPR loads that queue behind the staging loads.  With N = 256 the
// im2col form is the right one: the A operand re-read per tap comes from L2 (the hidden tensors are 11 MB), the tile is conv_fwd32d's
// (2 x 2 waves of 4 x 4 v_mfma_f32_16x16x4_f32 tiles, XOR-swizzled [row][32 k] images written by buffer_load ... lds), and a chunk
// is ONE tap x <= 8 channel quads of one segment, so the per-lane part of every address changes only when the walk moves to the next tap.
//   * Channel counts per tap are multiples of 4, not of 32: a tap's Q quads are split into ceil(Q / 8) balanced chunks (44 channels:
//     6 + 5 quads; 176: 8 + 8 + 7 + 7 + 7 + 7); the quads of a chunk beyond its length are not fetched (voffset beyond num_records).
//   * MFMA steps of a chunk: full sets of 4 quads through ds_read_b128 (lane quarter lq <-> quad 4h + lq, element t <-> step t, as in
//     conv_fwd32d), then the remaining 1..3 quads one step each through ds_read_b32 (lane quarter lq <-> element lq of quad 4h + t):
//     no zero-padded k-steps -- 44 channels cost 11 steps, not 12.
//   * Source pixels of a tap: per tile row two small LDS tables (row / column term of the 5 possible shifts, reflection or the
//     out-of-plane marker folded in), so moving to the next tap is 2 ds_read_b32 + 3 VALU per staged row.
#include "common.h"
#include <stdlib.h>

namespace cat_ks {

constexpr int BM = 128, MT = 4, WN = 2;      // N tile = WN * NT * 16 output channels: NT = 4 (128 wide) or 3 (96 wide: 176 channels = 2 tiles, 8 % padding instead of 31 %)
constexpr unsigned kOut = 0x80000000u;     // byte offset beyond every buffer: the buffer unit returns zeros and touches no memory
constexpr int kBadPix = 0x20000000;        // table marker of a source pixel in the zero padding / of a row beyond M

struct Seg {
  const float* src;
  const float* w;
  int xcs, q, ks, pad, wcs, wrow;          // q = channel quads per tap; wrow = floats per output channel of w (taps * wcs)
};
struct Args {
  Seg seg[CAT_KSUM_MAXSEG];
  int nseg;
  const float* bias;
  const float* res;
  float* y;
  int N, H, W, Cout, ycs, ycw, rcs, act, reflect, M;
  float slope;
  int var;      // DIAGNOSTIC BUILD ONLY (cat::kDiag; CAT_KSUM_VAR, results become wrong): 1 no epilogue stores, 4 no DMA after the first
                // chunk, 8 no remainder steps, 16 no residual read
};

__device__ __forceinline__ int swz32(int r, int q) { return (r * 8 + (q ^ ((r >> 1) & 7))) * 4; }

template <int NT>
__device__ __forceinline__ void ksum_body(const Args& p) {
  constexpr int BN = WN * NT * 16, BI = BN / 32;      // B rows of the tile, B staging instructions per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  float* sA = smem;
  float* sB = smem + 2 * BM * 32;
  int* tabY = reinterpret_cast<int*>(smem + 2 * (BM + BN) * 32);   // [128][8]: (n * H + reflect(oy + d)) * W for d = -2 .. 2, or kBadPix
  int* tabX = tabY + BM * 8;                                       // [128][8]: reflect(ox + d), or kBadPix
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.Cout + BN - 1) / BN;
  const int bid = cat::xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
  const int HW = p.H * p.W;
  const int var = cat::kDiag ? p.var : 0;

  // shift tables of the tile's 128 output pixels
  for (int e = tid; e < BM * 5; e += 256) {
    const int row = e / 5, d = e - row * 5 - 2;
    const int m = m0 + row;
    int ty = kBadPix, tx = kBadPix;
    if (m < p.M) {
      const int n = m / HW, rem = m - n * HW;
      const int oy = rem / p.W, ox = rem - oy * p.W;
      const int iy = oy + d, ix = ox + d;
      if (p.reflect) {
        ty = (n * p.H + cat::reflect_idx(iy, p.H)) * p.W;
        tx = cat::reflect_idx(ix, p.W);
      } else {
        if ((unsigned)iy < (unsigned)p.H) ty = (n * p.H + iy) * p.W;
        if ((unsigned)ix < (unsigned)p.W) tx = ix;
      }
    }
    tabY[row * 8 + d + 2] = ty;
    tabX[row * 8 + d + 2] = tx;
  }
  __syncthreads();

  // staging map (conv_fwd32d's): wave w, instruction i -> tile rows (w * 4 + i) * 8 + (lane >> 3), LDS slot lane & 7 <- source quad slot ^ swz(row).
  // (row >> 1) & 7 = (4 i + (lane >> 4)) & 7: instructions 0 / 2 fetch quad sqe, instructions 1 / 3 quad sqe ^ 4
  const int slot = lane & 7, lrow = lane >> 3;
  const int sqe = slot ^ ((lrow >> 1) & 3);
  int srow[4], srowB[BI], sqB[BI];
  unsigned voffC[4], voffN[4], voffB[BI];    // per staged row: source pixel of the walk's current / next tap (bytes), filter row (bytes)
#pragma unroll
  for (int i = 0; i < 4; ++i) srow[i] = (wave * 4 + i) * 8 + lrow;
#pragma unroll
  for (int i = 0; i < BI; ++i) {             // the B tile has BN = 32 BI rows: wave w, instruction i -> rows (w * BI + i) * 8 + (lane >> 3)
    srowB[i] = (wave * BI + i) * 8 + lrow;
    sqB[i] = slot ^ ((srowB[i] >> 1) & 7);
  }

  // walk state of the NEXT chunk to fetch (wave-uniform).  K of a segment = the flattened sequence of (tap, channel quad) pairs; a chunk is
  // the next <= 8 of them (q >= 8: it may straddle ONE tap boundary, each lane picks the tap its quad belongs to) or, for q < 8, one tap
  int ws = 0, wtap = 0, wq0 = 0;             // segment, tap of the chunk's first quad, that quad's index inside the tap
  int nky = 0, nkx = 0;                      // filter coordinates of tap wtap + 1 (the "next" tap, already located)
  int wq = 1, wks = 1, wtaps = 1, wpad = 0, wflat = 0;
  unsigned wxcs4 = 0, wwcs4 = 0;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].src), 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].w), 0, 0x7fffffff, 0x00020000);
  auto locate = [&](unsigned (&dst)[4], int ky, int kx) {      // branch-free: the eight table reads go out together
    const int dy = ky - wpad + 2, dx = kx - wpad + 2;
    int pix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pix[i] = tabY[srow[i] * 8 + dy] + tabX[srow[i] * 8 + dx];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned v = (unsigned)pix[i] * wxcs4;
      dst[i] = pix[i] < kBadPix ? v : kOut;
    }
  };
  auto step_next = [&]() {      // (nky, nkx) -> the tap after it
    if (++nkx == wks) {
      nkx = 0;
      ++nky;
    }
  };
  auto enter_segment = [&](int s) {
    const Seg& sg = p.seg[s];
    rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.src), 0, 0x7fffffff, 0x00020000);
    rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.w), 0, 0x7fffffff, 0x00020000);
    wq = sg.q;
    wflat = sg.q >= 8;
    wks = sg.ks;
    wtaps = sg.ks * sg.ks;
    wpad = sg.pad;
    wwcs4 = (unsigned)sg.wcs * 4u;
    wxcs4 = (unsigned)sg.xcs * 4u;
    const unsigned wrow4 = (unsigned)sg.wrow * 4u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int co = n0 + srowB[i];
      const unsigned v = (unsigned)co * wrow4;
      voffB[i] = co < p.Cout ? v : kOut;
    }
    wtap = wq0 = 0;
    nky = nkx = 0;
    locate(voffC, 0, 0);
    step_next();
    if (wtaps > 1) locate(voffN, nky, nkx);
  };
  enter_segment(0);
  // returns the quads of the chunk it issued
  auto issue = [&](int buf) -> int {
    const int left = (wtaps - wtap) * wq - wq0;             // quads of the segment not yet fetched
    const int nq = wflat ? min(8, left) : wq;
    if (!(var & 4)) {
      float* dA = sA + buf * BM * 32 + wave * 4 * 256;
      float* dB = sB + buf * BN * 32 + wave * BI * 256;
      const unsigned tapB = (unsigned)wtap * wwcs4;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int sq = sqe ^ (par * 4);                       // this lane's quad slot of the chunk in A instructions par, par + 2
        const int t = wq0 + sq;
        const bool nxt = t >= wq;                            // flat chunks only (q < 8: nq = q <= t is masked below)
        const unsigned qoff = (unsigned)(nxt ? t - wq : t) * 16u;
        const bool valid = sq < nq;
#pragma unroll
        for (int i = par; i < 4; i += 2) {
          const unsigned va = (nxt ? voffN[i] : voffC[i]) + qoff;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_t)(dA + i * 256), 16, valid ? va : kOut, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int t = wq0 + sqB[i];
        const bool nxt = t >= wq;
        const unsigned boff = (nxt ? tapB + wwcs4 : tapB) + (unsigned)(nxt ? t - wq : t) * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_t)(dB + i * 256), 16, sqB[i] < nq ? voffB[i] + boff : kOut, 0, 0, 0);
      }
    }
    wq0 += nq;
    if (wq0 >= wq) {          // the walk enters the next tap (wave-uniform)
      wq0 -= wq;
      if (++wtap == wtaps) {
        if (++ws < p.nseg) enter_segment(ws);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) voffC[i] = voffN[i];
        step_next();
        if (wtap + 1 < wtaps) locate(voffN, nky, nkx);
      }
    }
    return nq;
  };

  const int lr = lane & 15, lq = lane >> 4;
  // accumulators TRANSPOSED (filter fragment as the MFMA's first operand): lane (lr, lq) ends with output channels j * 16 + lq * 4 + 0..3 of
  // pixel i * 16 + lr -- one float4 per (i, j) in the epilogue instead of four scalar stores
  f4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  int arow[MT], brow[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) arow[i] = wm * MT * 16 + i * 16 + lr;
#pragma unroll
  for (int j = 0; j < NT; ++j) brow[j] = wn * NT * 16 + j * 16 + lr;
  auto mma = [&](int buf, int nq) {
    const float* A = sA + buf * BM * 32;
    const float* B = sB + buf * BN * 32;
    const int nfull = nq >> 2, rem = (var & 8) ? 0 : (nq & 3);
    for (int h = 0; h < nfull; ++h) {
      f4 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f4*>(A + swz32(arow[i], lq + 4 * h));
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f4*>(B + swz32(brow[j], lq + 4 * h));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][t], fa[i][t], acc[i][j], 0, 0, 0);
    }
    for (int r = 0; r < rem; ++r) {
      float ga[MT], gb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = A[swz32(arow[i], 4 * nfull + r) + lq];
#pragma unroll
      for (int j = 0; j < NT; ++j) gb[j] = B[swz32(brow[j], 4 * nfull + r) + lq];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[j], ga[i], acc[i][j], 0, 0, 0);
    }
  };

  int nq_cur = issue(0);
  __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0): the DMA has landed
  __syncthreads();
  int buf = 0;
  while (ws < p.nseg) {
    const int nq_next = issue(buf ^ 1);     // in flight behind this chunk's MFMA stream; buf ^ 1 was released by the barrier below
    mma(buf, nq_cur);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    buf ^= 1;
    nq_cur = nq_next;
  }
  // the residual quads of this lane are fetched behind the last chunk's MFMA stream (64 registers; the budget is 256 at two waves per SIMD):
  // read in the epilogue they were a round trip in front of the stores (teacher tail: 501 -> 494 us)
  f4 rr[MT][NT];
  {
    const bool rvec = p.res && (p.rcs & 3) == 0 && !(var & 16);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * MT * 16 + i * 16 + lr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c0 = n0 + wn * NT * 16 + j * 16 + lq * 4;
        rr[i][j] = (rvec && m < p.M && c0 + 3 < p.Cout) ? *reinterpret_cast<const f4*>(p.res + (int64_t)m * p.rcs + c0) : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  mma(buf, nq_cur);

  // epilogue: y = act(acc + bias) + res, four consecutive channels per lane
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * MT * 16 + i * 16 + lr;
    if (m >= p.M) continue;
    float* yo = p.y + (int64_t)m * p.ycs;
    const float* ro = p.res ? p.res + (int64_t)m * p.rcs : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c0 = n0 + wn * NT * 16 + j * 16 + lq * 4;
      if (c0 >= p.ycw) continue;
      f4 v = acc[i][j];
      if (c0 + 3 < p.Cout && (p.rcs & 3) == 0) {       // whole quad valid
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += p.bias[c0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cat::apply_act(v[e], p.act, p.slope);
        v += rr[i][j];
        if (!(var & 1) || v[0] == 123.456f) *reinterpret_cast<f4*>(yo + c0) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + e;
          if (c >= p.ycw) continue;
          float o = 0.f;
          if (c < p.Cout) {
            o = cat::apply_act(v[e] + (p.bias ? p.bias[c] : 0.f), p.act, p.slope);
            if (ro) o += ro[c];
          }
          yo[c] = o;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ksum_kernel4(const Args p) { ksum_body<4>(p); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ksum_kernel3(const Args p) { ksum_body<3>(p); }

}  // namespace cat_ks

extern "C" {

int cat_conv2d_ksum_supported(const cat_ksum_t* g) {
  if (g->nseg < 1 || g->nseg > CAT_KSUM_MAXSEG || g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Cout <= 0) return 0;
  int refl = -1;
  for (int s = 0; s < g->nseg; ++s) {
    const cat_ksum_seg_t& sg = g->seg[s];
    if (!(sg.ks == 1 || sg.ks == 3 || sg.ks == 5) || sg.c4 <= 0 || (sg.c4 & 3) || (sg.xcs & 3) || sg.xcs < sg.c4 || (sg.wcs & 3) || sg.wcs < sg.c4) return 0;
    if ((int64_t)g->N * g->H * g->W * sg.xcs * 4 >= (int64_t)2147483647 || (int64_t)g->Cout * sg.ks * sg.ks * sg.wcs * 4 >= (int64_t)2147483647) return 0;
    if (sg.ks > 1) {
      if (refl >= 0 && refl != (sg.reflect != 0)) return 0;      // one padding mode per launch (one pair of shift tables)
      refl = sg.reflect != 0;
      if (sg.reflect && (sg.ks / 2 >= g->H || sg.ks / 2 >= g->W)) return 0;
    }
  }
  return 1;
}

int cat_conv2d_ksum_fwd(const cat_ksum_t* g, const float* bias, const float* res, float* y, cat_stream_t stream) {
  CAT_REQUIRE(cat_conv2d_ksum_supported(g), "conv ksum: unsupported geometry (stride-1 same convs, k in {1,3,5}, channel counts / strides multiples of 4, "
                                            "one padding mode, tensors < 2 GB)");
  CAT_REQUIRE((g->ycs & 3) == 0 && g->ycs >= g->Cout && g->ycw <= g->ycs, "conv ksum: bad output stride");
  CAT_REQUIRE(res == nullptr || g->rcs >= g->Cout, "conv ksum: residual stride");
  cat_ks::Args a{};
  double kflops = 0.0;
  a.reflect = 0;
  for (int s = 0; s < g->nseg; ++s) {
    const cat_ksum_seg_t& sg = g->seg[s];
    cat_ks::Seg& d = a.seg[s];
    d.src = sg.src; d.w = sg.w; d.xcs = sg.xcs; d.q = sg.c4 >> 2; d.ks = sg.ks; d.pad = sg.ks >> 1; d.wcs = sg.wcs; d.wrow = sg.ks * sg.ks * sg.wcs;
    if (sg.ks > 1 && sg.reflect) a.reflect = 1;
    kflops += (double)sg.ks * sg.ks * (sg.cin > 0 ? sg.cin : sg.c4);
  }
  a.nseg = g->nseg; a.bias = bias; a.res = res; a.y = y;
  a.N = g->N; a.H = g->H; a.W = g->W; a.Cout = g->Cout; a.ycs = g->ycs; a.ycw = g->ycw > g->Cout ? g->ycw : g->Cout; a.rcs = g->rcs;
  a.act = g->act; a.slope = g->slope;
  const int64_t M = (int64_t)g->N * g->H * g->W;
  CAT_REQUIRE(M < (int64_t)cat_ks::kBadPix, "conv ksum: too many pixels");
  a.M = (int)M;
  // 96-wide N tiles where they pad less than 128-wide ones (176 output channels: 192 instead of 256 columns)
  const bool n96 = cat::cdiv(g->Cout, 96) * 96 < cat::cdiv(g->Cout, 128) * 128;
  const int bn = n96 ? 96 : 128;
  const int64_t grid = (int64_t)cat::cdiv(M, cat_ks::BM) * cat::cdiv(g->Cout, bn);
  CAT_REQUIRE(grid < (int64_t)2147483647, "conv ksum: grid too large");
  const size_t lds = (size_t)2 * (cat_ks::BM + bn) * 32 * sizeof(float) + (size_t)2 * cat_ks::BM * 8 * sizeof(int);
  static cat::LdsOptIn optin3, optin4;
  if (n96) cat::lds_optin(optin3, (const void*)cat_ks::ksum_kernel3, (int)lds);
  else cat::lds_optin(optin4, (const void*)cat_ks::ksum_kernel4, (int)lds);
  static const int var_env = [] {
    const int v = (cat::kDiag && getenv("CAT_KSUM_VAR")) ? atoi(getenv("CAT_KSUM_VAR")) : 0;
    if (v) fprintf(stderr, "libcat_hip: CAT_KSUM_VAR=%d -- ksum results are INTENTIONALLY WRONG (timing diagnostics only)\n", v);
    return v;
  }();
  a.var = var_env;
  cat::ProfScope prof("conv_ksum", 2.0 * (double)M * g->Cout * kflops, 0.0, stream);
  if (n96) cat_ks::ksum_kernel3<<<(int)grid, 256, lds, (hipStream_t)stream>>>(a);
  else cat_ks::ksum_kernel4<<<(int)grid, 256, lds, (hipStream_t)stream>>>(a);
  return cat::check_launch("conv2d_ksum_fwd");
}

}  // extern "C"
