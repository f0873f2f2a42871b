// Shared device/host helpers for libcat_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cat_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));

namespace cat {

// Timing diagnostics -- per-phase shader clocks (CAT_DBG), scheduling experiments (CAT_SCHED, CAT_LDS_PAD) and the ablation switches that make
// results WRONG by design (CAT_PK_ABLATE, CAT_Q_ABLATE) -- exist only in the DIAGNOSTIC build of the library: `python -m cat_amd._build --diag`
// compiles with -DCAT_DIAG into lib/libcat_hip_diag.so, which tools/debug/* load through CAT_LIB=diag.  In the production library every
// `kDiag && ...` condition is a compile-time false: the branches and their operands are not in the kernels.
#ifdef CAT_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Optional HIP-event timing of one entry-point call (enabled by cat_prof_enable): records an event pair on the
// launch stream around everything the enclosing scope enqueues, tagged with its algorithmic FLOPs / bytes.
class ProfScope {
 public:
  ProfScope(const char* family, double flops, double bytes, void* stream);
  ~ProfScope();

 private:
  bool active_;
  void* stream_;
  int idx_;
};

#define CAT_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      cat::set_error(__VA_ARGS__);        \
      return -22;                         \
    }                                     \
  } while (0)

// Cout <= 4 direct convolutions (conv_smallco.hip)
bool smallco_applicable(const cat_conv_t* g);
int smallco_fwd_ksplit(const cat_conv_t* g);   // > 1: channel-split forward, needs ksplit * N*Ho*Wo*ycs floats of workspace
int smallco_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, float* ws, int ksplit, hipStream_t s);
int smallco_wgrad_nblk(const cat_conv_t* g);
int smallco_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s);
bool smallci_dgrad_applicable(const cat_conv_t* g);   // 4x4 / stride 2 / pad 1 conv with 3 or 6 input channels (PatchGAN's first layer)
int smallci_dgrad(const cat_conv_t* g, const float* dy, const float* w, float* dx, int dxcs, int dxcw, hipStream_t s);

// LDS-tile weight gradient of narrow stride-1 3x3 / 5x5 layers (conv_twgrad.hip)
bool twgrad_applicable(const cat_conv_t* g);
int twgrad_nblk(const cat_conv_t* g);
int twgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s);   // partials [nblk][Cout][taps * round_up(Cin, 4)]

// pixel-streaming weight gradient of 1 x 1 layers with few channels and many pixels (conv_pwgrad.hip)
bool pwgrad_applicable(const cat_conv_t* g);
int pwgrad_nblk(const cat_conv_t* g);
int pwgrad(const cat_conv_t* g, const float* x, const float* dy, float* ws, hipStream_t s);   // partials [nblk][Cout][round_up(Cin, 4)]

// Opt a kernel in to more than 64 KB of dynamic LDS.  hipFuncSetAttribute applies to the CURRENT device only, so the "already
// done" flag of a call site is kept per device ordinal (a process may drive several GPUs, e.g. the 2-ranks-on-one-box tests).
struct LdsOptIn {
  unsigned long long done = 0ull;
};
static inline void lds_optin(LdsOptIn& st, const void* fn, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && ((st.done >> dev) & 1ull)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0 && dev < 64) st.done |= 1ull << dev;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case CAT_ACT_RELU: return v > 0.f ? v : 0.f;
    case CAT_ACT_LRELU: return v > 0.f ? v : v * slope;
    case CAT_ACT_TANH: return tanhf(v);
    case CAT_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    default: return v;
  }
}

// derivative of the activation expressed through its OUTPUT y (all four are invertible enough for that)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  switch (act) {
    case CAT_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case CAT_ACT_LRELU: return y > 0.f ? 1.f : slope;
    case CAT_ACT_TANH: return 1.f - y * y;
    case CAT_ACT_RELU6: return (y > 0.f && y < 6.f) ? 1.f : 0.f;      // hardtanh_backward: 0 at and beyond both bounds
    default: return 1.f;
  }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// XCD-aware, bijective block remap (guide §5 T1): consecutive logical tiles land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace cat
