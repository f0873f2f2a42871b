/* Opt-in entry points of libcat_hip.so that are NOT part of the default (exact fp32) path: kept in their own header so that cat_hip.h describes exactly
 * what the graded configuration runs.  Plain C ABI, same conventions as cat_hip.h. */
#ifndef CAT_HIP_SPLIT_H
#define CAT_HIP_SPLIT_H
#include "cat_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Opt-in split-bf16 ("bf16x3") forms of the two wide backward tiles (csrc/conv_split.hip; switch CAT_MFMA=bf16x3 on the host side, never the default):
 * NLayerDiscriminator's 4 x 4 convolutions (models/modules/discriminators.py:38-76) whose data / weight gradients cat_conv2d_dgrad_t / cat_conv2d_wgrad
 * compute in exact fp32.  Operands arrive PRE-SPLIT, x = x1 + x2 with x1 = bf16(x), x2 = bf16(x - x1): planes [2][n] of 16-bit values
 * (cat_split_bf16); a product is x1 y1 + x1 y2 + x2 y1 on the bf16 matrix pipe with fp32 accumulation (<= 2^-16 of a product dropped).
 * Eligibility (…_applicable): zero padding, 4 x 4 kernel at stride 1 | 2, Cin % 128 == 0, dense activations; dgrad: Cout % 64 == 0;
 * wgrad: Cout % 128 == 0, Wo <= 64.  wt_planes = the split of cat_conv2d_weight_transpose's [Cin][kh][kw][Cout] copy. */
int cat_split_bf16(const float* x, void* planes, int64_t n, cat_stream_t stream);
int cat_conv2d_dgrad_split_applicable(const cat_conv_t* g);
int cat_conv2d_dgrad_split(const cat_conv_t* g, const void* dy_planes, const void* wt_planes, float* dx, int dxcs, cat_stream_t stream);
int cat_conv2d_wgrad_split_applicable(const cat_conv_t* g);
size_t cat_conv2d_wgrad_split_ws_bytes(const cat_conv_t* g);
int cat_conv2d_wgrad_split(const cat_conv_t* g, const void* x_planes, const void* dy_planes, float* dw, int accumulate, void* ws,
                           cat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
