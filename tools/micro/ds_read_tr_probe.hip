// Probe (round 4): which (lane, element) does each result element of gfx950's ds_read_b64_tr_b16 come from?  Every lane passes the address of ITS four
// contiguous 16-bit values (lane l: elements 4 l .. 4 l + 3 of an LDS array holding its own indices), so a returned value v names source lane v / 4 and
// source element v % 4.  Needed for the weight-gradient form of the split-bf16 tiles, whose K dimension (pixels) is the slow one of both operands.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ds_read_tr_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s4* lp;
  const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
  short* d;
  (void)hipMalloc(&d, 256 * 2);
  probe<<<1, 64>>>(d);
  short h[256];
  (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (lane %2d, elem %d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
