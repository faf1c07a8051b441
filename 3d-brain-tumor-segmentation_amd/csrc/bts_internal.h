// Internal constants shared by the .hip translation units; the public C ABI is include/bts_hip.h.
#pragma once
#include "../../include/bts_hip.h"

// groupnorm.hip: mean / rstd from per-block (sum, sumsq) partials laid out [N*G][B][2] (slab semantics), fixed order
int bts_gn_finalize_partials_(const double* partial, float* mean, float* rstd, int NG, long B, double count, float eps,
                              hipStream_t stream);

// groupnorm.hip, for block_bwd.hip: blocks per (sample, slab) unit / elements per block of the vectorised slab kernels (false: generic shape)
bool bts_gn_slab_blocks_(int N, long V, int C, int G, int* B, long* span);

// se.hip: stage 2 of the gate backward alone (SE-MLP backward from the summed partials red[(n*F+c)*2 + {ch, w}])
int bts_se_mlp_bwd_(const double* red, double* scratch, const float* gap, const float* h, const float* ch, const float* w1, const float* w2,
                    float* dw1, float* dw2, float* dwsp, float* dgap, int N, int B, long V, int F, int R, int accumulate_params,
                    hipStream_t stream);

// conv_wino.hip: 3x3x3 stride-1 conv in Winograd F(2x2,3x3) x direct form; BTS_OK = taken, 1 = declined (run the implicit GEMM)
int bts_wino_launch_(const float* x, const float* up, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                     int Cout, int ldy, int accum, double* gnp, int gnG, long* gn_B, void* ws, long ws_bytes, hipStream_t stream);
long bts_wino_workspace_(int N, int D, int H, int W, int Cin, int Cout);
// conv_wino3.hip: the same convolution in Winograd F(2x2x2,3x3x3) form on the third part of the K3S1 image; offered first
int bts_w3_launch_(const float* x, const float* up3, const float* bias, float* y, int N, int D, int H, int W, int Cin, int ldx,
                   int Cout, int ldy, int accum, double* gnp, int gnG, long* gn_B, void* ws, long ws_bytes, hipStream_t stream);
long bts_w3_workspace_(int N, int D, int H, int W, int Cin, int Cout);
// conv_igemm.hip: y (+)= bias + sum_z part[z][voxel][Npad]  (finish of a split-K launch, fixed summation order)
int bts_igemm_reduce_(const float* part, const float* bias, float* y, long nvox, int Cout, int Npad, int ldy, int ksplit,
                      int with_bias, int accum, hipStream_t stream);
// se.hip: stages 2a + 2 of the gate backward (sum the per-block partials [n][b][F][{ch, w}], SE-MLP backward, dgap / V)
int bts_se_mlp_bwd_sample_(const double* red, double* scratch, const float* h, const float* ch, const float* w1, const float* w2, float* dgap, int N,
                           long V, int F, int R, hipStream_t stream);
int bts_se_bwd_middle_(double* partial, double* red, double* scratch, const float* gap, const float* h, const float* ch, const float* w1,
                       const float* w2, float* dw1, float* dw2, float* dwsp, float* dgap, int N, int B, long V, int F, int R,
                       int accumulate_params, hipStream_t stream);

// conv_igemm.hip: a kernel is about to read part `bit` (1 implicit-GEMM, 2 F(2x2,3x3) x direct, 4 F(2x2x2,3x3x3)) of the K3S1 weight image
// that starts at `base`: recorded (re-packs then write the parts in use only), and packed on the spot where the last re-pack left it out
int bts_img_note_use_(const float* base, unsigned bit, hipStream_t stream);
int bts_img_ensure_(const float* base, unsigned bit, hipStream_t stream);
void bts_img_mark_used_(const float* base, unsigned bit);
