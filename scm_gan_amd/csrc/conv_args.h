// Kernel-argument block shared by the 3x3 MFMA conv kernels (conv3x3_mfma.hip, conv3x3_m16.hip).
#pragma once
#include "common.h"

#define KC 8
#define MAXJ 8

struct ConvKArgs {
    const float* x1; const float* x2; int c1; int c2;
    const float* gate1; const float* gate2; int gate_act;
    const float* wpack; int cout;
    const float* scale; const float* bias; const float* sbias;
    float* y1; float* y2; int co1;
    int B; int H; int W;
    int pad_mode; int act;
    int tiles_per_img; int plane;
    const unsigned* mask_in; unsigned* mask_out;   // 1-bit leaky-relu gate, words [B][ceil(C/32)][HW] (scm_hip.h)
    int acc_y1;                                    // y1 += result (a second consumer's input gradient lands in the same buffer)
    const unsigned* ymask1; const unsigned* ymask2; // conv3x3_wino.hip: leaky-relu gates applied to what is written to y1 / y2 (scm_hip.h)
    int x_half = 0, y_half = 0;                    // conv3x3_h16.hip: x1 / x2 (y1 / y2) are IEEE-half tensors (scm_hip.h)
};


// cout <= 16 variant on v_mfma_f32_16x16x4_f32 (conv3x3_m16.hip)
int scm_launch_conv3x3_m16(const ConvKArgs& k, hipStream_t s);

// cout <= 16 Winograd variant on v_mfma_f32_16x16x4_f32 (conv3x3_wino16.hip): pack size / builder, frame check, launch
size_t scm_wino16_pack_floats(int cin);
int scm_pack_wino16(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0, float* fwd_pack,
                    float* dgrad_pack, hipStream_t s);
bool scm_wino16_serves(int H, int W);
int scm_launch_wino16(const ConvKArgs& k, hipStream_t s);

// <= 16 INPUT channels, many output channels (conv3x3_wino_k16.hip): its pack follows the wide kernel's in the same buffer
size_t scm_wino_k16_pack_floats(int n_out);
int scm_pack_wino_k16(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0, float* fwd_pack,
                      float* dgrad_pack, hipStream_t s);
bool scm_wino_k16_serves(const ConvKArgs& a);
int scm_launch_wino_k16(const ConvKArgs& k, hipStream_t s);

// Workgroup -> (image, tile, output-channel group) for grids of (B * tiles_per_img, groups): neighbouring tiles of ONE
// image (they share halo rows) and the channel groups of a tile (they read the same rows) run on ONE XCD, next to each
// other in time -- each XCD has its own L2 and the hardware deals consecutive workgroup ids (x fastest) round robin over
// the 8 XCDs, so with the plain mapping every shared row is fetched from HBM once per XCD that touches it.
__device__ __forceinline__ void scm_xcd_tile_map(int B, int tiles_per_img, int& b, int& wt, int& cg) {
    const int groups = gridDim.y;
    if ((B & 7) == 0) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        const int q = id >> 3;
        cg = q % groups;
        const int r = q / groups;
        wt = r % tiles_per_img;
        b = (r / tiles_per_img) * 8 + (id & 7);
    } else {
        b = blockIdx.x / tiles_per_img;
        wt = blockIdx.x - b * tiles_per_img;
        cg = blockIdx.y;
    }
}

// two tiles per workgroup (conv3x3_wino_t2.hip): the hot 128-channel instantiation on 64-wide frames
bool scm_wino_t2_serves(const ConvKArgs& k);
int scm_launch_wino_t2(const ConvKArgs& k, int groups, hipStream_t s);
