// Kernel-argument block and work split shared by the weight-gradient kernels
// (conv3x3_wgrad.hip, conv3x3_wgrad_h16.hip).
#pragma once
#include "common.h"

struct WgradKArgs {
    const float* x1; const float* x2; int c1; int c2;
    const float* dy; const float* gate; int gate_act; int cout;
    float* part; float* csum_part;
    int B; int H; int W; int pad_mode;
    int spi; int rows_per_split; int nseg; int PS;
    int ci_groups;
    const unsigned* gate_mask;     // 1-bit leaky-relu gate, words [B][ceil(cout/32)][HW], or null
    int x_half = 0;                // x1 / x2 are IEEE-half tensors (conv3x3_wgrad_h16.hip, the direct 16-channel kernels)
};

// Workgroup -> (split, channel group) for grids of (B * spi splits, groups): the groups of one split read the same rows
// and the splits of one image share border rows / columns, so one IMAGE runs on ONE XCD (own L2), its workgroups next
// to each other in time.  The hardware deals consecutive workgroup ids (x fastest) round robin over the 8 XCDs; with
// the plain mapping the groups of a split start a whole grid row apart and each of them fetches its rows from HBM
// again (PMC: 718 MB per conv6 weight gradient, 4.7 TB/s, against 290 MB of distinct bytes).
__device__ __forceinline__ void scm_xcd_split_group(int spi, int& split, int& group) {
    const int nsplit = gridDim.x, groups = gridDim.y;
    const int id = blockIdx.y * nsplit + blockIdx.x;
    if ((nsplit & 7) == 0 && ((nsplit / spi) & 7) == 0) {
        const int q = id >> 3;
        group = q % groups;
        const int r = q / groups;
        split = ((r / spi) * 8 + (id & 7)) * spi + r % spi;
    } else {
        split = blockIdx.x;
        group = blockIdx.y;
    }
}

struct WgradPlan {
    int mt, nt, spi, rows_per_split, nseg, PS, co_groups, ci_groups, nsplit, m16;
    int ws_splits;      // partial slabs the workspaces are sized for (>= nsplit: the thin Winograd form splits a row into
                        // 32-column parts and needs B * ceil(W / 32) of them)
};

// 128 co x 64 ci tiles on v_mfma_f32_32x32x16_{f16,bf16} (conv3x3_wgrad_h16.hip)
int scm_launch_wgrad_h16(const WgradKArgs& a, const WgradPlan& p, int precision, hipStream_t s);

// fp32 arithmetic on the bf16 MFMA pipe, 128 co x 32 ci tiles (conv3x3_wgrad_x3.hip)
int scm_launch_wgrad_x3(const WgradKArgs& a, const WgradPlan& p, hipStream_t s);

// Winograd F(2x2,3x3) weight gradient, 64 co x 64 ci tiles on the f32 MFMA pipe (conv3x3_wgrad_wino.hip): not a
// precision -- an algorithm tag that travels in wgrad_impl's `precision` argument.  Same slabs, its own split.
#define SCM_ALGO_WINO 16
bool scm_wgrad_wino_serves(const WgradKArgs& a);
int scm_launch_wgrad_wino(const WgradKArgs& a, int max_splits, int* nsplit_out, int* spi_out, hipStream_t s);

// Winograd F(2x2,3x3) weight gradient of the layers with at most 16 channels on one side, 16 x 16 tiles on
// v_mfma_f32_16x16x4_f32 with operands transformed in the lanes that feed them (conv3x3_wgrad_wino16.hip).
bool scm_wgrad_wino16_serves(const WgradKArgs& a, int max_splits);
int scm_launch_wgrad_wino16(const WgradKArgs& a, int max_splits, int* nsplit_out, int* spi_out, hipStream_t s);

// The same tiles with the transforms written around the register pairs the loads deliver and a chunk walk that visits
// both 64-byte halves of a line back to back (conv3x3_wgrad_winoz.hip); takes the first form's split as it is.
bool scm_wgrad_winoz_serves(const WgradKArgs& a);
int scm_launch_wgrad_winoz(const WgradKArgs& a, dim3 grid, hipStream_t s);
