// Launch functions of the software-pipelined bf16x3 kernels (conv_x3.hip), called by conv.hip / conv_winograd.h.
#pragma once
#include "lmh_common.h"
int lmh_x3_fwd_launch(const lmh_conv_desc* d, const float* x, const float* w, const float* scale, const float* shift,
                      const float* residual, float* y, uint32_t* act_bits, int gbatch, int bm, int bn, int pipe, hipStream_t st);
int lmh_x3_bwd_data_launch(const lmh_conv_desc* d, const float* dy, const float* w, const float* kscale,
                           const float* addend, const uint32_t* xbits, float* dx, int bm, int bn, int pipe, hipStream_t st);
int lmh_x3_bwd_weight_launch(const lmh_conv_desc* d, const float* x, const float* g, float* out, int kt_per_split,
                             int tiles_x, int tiles_y, int splits, float* colpart, bool gb, int bm, int bn,
                             int pipe, hipStream_t st);
bool lmh_x3_bwd_weight_plain(const lmh_conv_desc* d, bool gb, int pipe);   // the PLAIN instantiation is what the launch above takes
// pipe: 0 = the round-2 schedule (one LDS buffer, two blocks per CU), 1 = software-pipelined (conv_x3.h)
// pre-split weights (round 6): bytes of a layer's W3 planes; one launch that splits n layers; forward with them
size_t lmh_x3_w3_bytes(int rs, int c, int k, int fwd);
int lmh_x3_split_launch(const float* const* w, void* const* out, const int* rs, const int* c, const int* k, int n, int fwd,
                        hipStream_t st);
int lmh_x3_fwd_ws_launch(const lmh_conv_desc* d, const float* x, const void* w3, const float* scale, const float* shift,
                         const float* residual, float* y, uint32_t* act_bits, int gbatch, int bm, int bn, hipStream_t st);
int lmh_x3_bwd_data_ws_launch(const lmh_conv_desc* d, const float* dy, const void* w3, const float* kscale,
                              const float* addend, const uint32_t* xbits, float* dx, int bm, int bn, hipStream_t st);
