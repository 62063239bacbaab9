// Fused decoder + pixel loss: ONE pass over the latent z.
//
// Replaces Decoder.forward (models.py:270-291: ConvTranspose 16->64 + leaky_relu, ConvTranspose
// 64->16*C, view(B,16,C,H,W).sum(1)) followed by torch.sigmoid + decoder_pixel_loss (main.py:188-190,
// 310-312) / the squared error of the eval rollout (main.py:814-816), and their autograd backward.
// The 64-channel hidden map and the 48-channel per-factor map never exist in HBM: per frame the
// forward reads 16 latent planes and writes C logit planes (311 296 B at 64x64, C=3).
//
// Algebra (exact): a stride-1 pad-1 ConvTranspose2d is the zero-padded 3x3 conv with channel-transposed,
// tap-flipped weights, and the factor sum of a linear conv is the conv with the factor-summed weights:
//   h[co]   = lrelu(b1[co] + sum_{ci,ky,kx} W1c[co][ci][ky][kx] z[ci](y+ky-1, x+kx-1)),  W1c = Wt1[ci][co][2-ky][2-kx]
//   l[c]    = b2s[c] + sum_{ch,ky,kx} W2c[c][ch][ky][kx] h[ch](y+ky-1, x+kx-1),  W2c = sum_f Wt2[ch][f*C+c][2-ky][2-kx]
// h is zero outside the image (conv2's zero padding), NOT the conv of out-of-image zeros.
//
// Forward kernel, one workgroup (4 waves) per band of TH image rows x 64 columns, row by row:
//   GEMM1  D1[co][px] over K = 16 ci x 9 taps on v_mfma_f32_32x32x2_f32: A = weights (lane -> co), B = pixels of the
//          z rows staged in LDS (im2col = an address offset); wave (mt, g) owns 32 co x 32 px.
//   GEMM2  in "scatter form": P[(c,tap)][px] = sum_ch W2c[c][ch][tap] h[ch][px].  D1's accumulator registers ARE the
//          B operand of this product (32x32 C/D layout: column on the lane, rows in the registers), so the hidden map
//          goes from one MFMA to the next without leaving the register file; the k order of the weight pack follows
//          the accumulator's row order.  27 of 32 rows used (no padding of C=3 outputs to an MFMA tile).
//   logits l[c](y, x) = b2s[c] + sum_{ky,kx} P[(c,ky,kx)](y+ky-1, x+kx-1): P of one h row goes through LDS and is added
//          into three rolling per-thread row accumulators in a fixed order (deterministic).
//   loss   sigmoid + BCE (log clamp -100) or squared error against the target row, per-workgroup partial sums,
//          reduced per sample in a fixed order by a second tiny kernel.
// Frames wider than 64 use overlapping 64-column tiles (62 owned output columns each).
#include "common.h"

#define DF_LAT 16
#define DF_HID 64
#define DF_ZS 72            // floats per staged z row (66 used: 64 columns + 1 halo each side)
#define DF_NZ (DF_LAT * 66)
#define DF_ZL ((DF_NZ + 255) / 256)

// pack layout (floats)
#define DF_OFF_W1P 0                       // [72 k-steps][2 mt][64 lanes]   GEMM1 A operand
#define DF_OFF_W2P 9216                    // [2 mt][16 regs][64 lanes]      GEMM2 A operand (k order = accumulator rows)
#define DF_OFF_B1 11264                    // [64]
#define DF_OFF_B2 11328                    // [4]  factor-summed conv2 bias
#define DF_FWD_PACK 11332
#define DF_OFF_W1Q 11392                   // [2 mt][5 tiles][16 regs][64 lanes]   backward: Q = W1c^T dh   A operand
#define DF_OFF_W2D 21632                   // [2 mt][14 k-steps][64 lanes]   backward: dh = W2c^T dl   A operand
#define DF_PACK_FLOATS 23424

extern "C" size_t scm_decoder_fused_pack_floats(void) { return DF_PACK_FLOATS; }

// accumulator register j of lane half hh holds row (j&3) + 8*(j>>2) + 4*hh of a 32x32 tile
__device__ __forceinline__ int acc_row(int j, int hh) { return (j & 3) + 8 * (j >> 2) + 4 * hh; }

__device__ __forceinline__ float w1c(const float* __restrict__ wt1, int co, int ci, int ky, int kx) {
    return wt1[((ci * DF_HID + co) * 3 + (2 - ky)) * 3 + (2 - kx)];       // Wt1 (16, 64, 3, 3)
}
__device__ __forceinline__ float w2c(const float* __restrict__ wt2, int L, int C, int c, int ch, int ky, int kx) {
    float s = 0.f;
    for (int f = 0; f < L; ++f) s += wt2[((ch * (L * C) + f * C + c) * 3 + (2 - ky)) * 3 + (2 - kx)];   // Wt2 (64, L*C, 3, 3)
    return s;
}

__global__ void decoder_fused_pack_kernel(const float* __restrict__ wt1, const float* __restrict__ b1,
                                          const float* __restrict__ wt2, const float* __restrict__ b2, int L, int C,
                                          float* __restrict__ pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DF_PACK_FLOATS) return;
    float v = 0.f;
    if (i < DF_OFF_W2P) {                                       // w1p[s][mt][l]
        const int l = i & 63, mt = (i >> 6) & 1, s = i >> 7;
        const int cp = s / 9, tap = s - cp * 9;
        v = w1c(wt1, mt * 32 + (l & 31), 2 * cp + (l >> 5), tap / 3, tap % 3);
    } else if (i < DF_OFF_B1) {                                 // w2p[mt][j][l]
        const int e = i - DF_OFF_W2P, l = e & 63, j = (e >> 6) & 15, mt = e >> 10;
        const int ct = l & 31, ch = mt * 32 + acc_row(j, l >> 5);
        if (ct < 9 * C) v = w2c(wt2, L, C, ct / 9, ch, (ct % 9) / 3, ct % 3);
    } else if (i < DF_OFF_B2) {
        v = b1[i - DF_OFF_B1];
    } else if (i < DF_FWD_PACK) {
        const int c = i - DF_OFF_B2;
        if (c < C) for (int f = 0; f < L; ++f) v += b2[f * C + c];
    } else if (i >= DF_OFF_W1Q && i < DF_OFF_W2D) {             // w1q[mt][t][j][l]: A operand of Q = W1c^T dh
        // Q[m][px] = sum_co A[m][co] dh[co][px]; the B operand is wave (mt, .)'s dh accumulator, register j of
        // lane half hh = channel co = mt*32 + acc_row(j, hh).  Output row m = t*32 + (l&31) of tile t lands in lane
        // half hm = (m>>2)&1, register i = ((m&31)>>3)*4 + (m&3); per half the 80 (tile, register) slots jj = t*16+i
        // carry (ci, tap) = (8*hm + jj/9, jj%9), jj < 72: every lane owns all 9 taps of its 8 input channels.
        const int e = i - DF_OFF_W1Q, l = e & 63, j = (e >> 6) & 15, t = (e >> 10) % 5, mt = e / 5120;
        const int mm = l & 31, hm = (mm >> 2) & 1, reg = (mm >> 3) * 4 + (mm & 3), jj = t * 16 + reg;
        if (jj < 72) {
            const int ci = 8 * hm + jj / 9, tap = jj % 9;
            v = w1c(wt1, mt * 32 + acc_row(j, l >> 5), ci, tap / 3, tap % 3);
        }
    } else if (i >= DF_OFF_W2D) {                               // w2d[mt][s][l]: row m = ch, k = (c,tap) = 2*s + (l>>5)
        const int e = i - DF_OFF_W2D, l = e & 63, s = (e >> 6) % 14, mt = e / (64 * 14);
        const int ct = 2 * s + (l >> 5);
        if (ct < 9 * C) v = w2c(wt2, L, C, ct / 9, mt * 32 + (l & 31), (ct % 9) / 3, ct % 3);
    }
    pack[i] = v;
}

extern "C" int scm_decoder_fused_pack(const float* conv1_weight, const float* conv1_bias, const float* conv2_weight,
                                      const float* conv2_bias, int latent, int C, float* pack, void* stream) {
    SCM_REQUIRE(conv1_weight && conv1_bias && conv2_weight && conv2_bias && pack, "scm_decoder_fused_pack: null pointer");
    SCM_REQUIRE(latent == DF_LAT && C >= 1 && C <= 3,
                "scm_decoder_fused_pack: the fused decoder serves latent_size 16 and 1..3 colour channels (got %d, %d)",
                latent, C);
    hipLaunchKernelGGL(decoder_fused_pack_kernel, dim3((DF_PACK_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       conv1_weight, conv1_bias, conv2_weight, conv2_bias, latent, C, pack);
    SCM_LAUNCH_CHECK("decoder_fused_pack_kernel");
    return SCM_OK;
}

// ------------------------------------------------------------------------------------------ tiling
struct DecGeom {
    int B, H, W, C, TH, nbands, nxt;
};
__host__ __device__ static inline int df_c0(int xt, int nxt, int W) {
    if (nxt == 1) return 0;
    const int c = xt * 62;
    return c < W - 64 ? c : W - 64;
}
// owned output columns [lo, hi) of x-tile xt
__host__ __device__ static inline void df_own(int xt, int nxt, int W, int* lo, int* hi) {
    if (nxt == 1) { *lo = 0; *hi = W; return; }
    const int c0 = df_c0(xt, nxt, W);
    *hi = (xt == nxt - 1) ? W : c0 + 63;
    if (xt == 0) { *lo = 0; return; }
    const int ph = df_c0(xt - 1, nxt, W) + 63;
    *lo = c0 + 1 > ph ? c0 + 1 : ph;
}
static void df_geom(int B, int H, int W, int C, DecGeom* g) {
    g->B = B; g->H = H; g->W = W; g->C = C;
    g->nxt = W <= 64 ? 1 : (W - 2 + 61) / 62;
    // rows per band: as tall as possible (the two halo rows are recomputed per band) while the grid still
    // fills the chip (2 workgroups per CU x 256 CUs)
    int th = 32;
    while (th > 4 && (long)B * g->nxt * ((H + th - 1) / th) < 512) th >>= 1;
    if (th > H) th = H;
    g->TH = th;
    g->nbands = (H + th - 1) / th;
}

// per-workgroup partial of the weight gradients: dW1 [64 co][160] (column ci*9+tap, column 144 = db1),
// dW2 [64 ch][32] (column c*9+tap), db2 [4]
#define DF_BWD_PART (64 * 160 + 64 * 32 + 4)
static void df_geom_bwd(int B, int H, int W, int C, DecGeom* g) {
    g->B = B; g->H = H; g->W = W; g->C = C;
    g->nxt = W <= 64 ? 1 : (W - 2 + 61) / 62;
    // one workgroup per CU (the kernel keeps ~400 VGPRs per lane and 120 KB of LDS): bands as tall as a grid of
    // >= 256 workgroups allows, which also bounds the partial slabs the reduce kernel reads
    int th = 64;
    while (th > 4 && (long)B * g->nxt * ((H + th - 1) / th) < 256) th >>= 1;
    if (getenv("SCM_DF_TH")) th = atoi(getenv("SCM_DF_TH"));
    if (th > H) th = H;
    g->TH = th;
    g->nbands = (H + th - 1) / th;
}

// ----------------------------------------------------------------------------------------- forward
struct DecFwdArgs {
    const float* z; const float* pack; const float* target; float* logits; float* loss_part;
    DecGeom g; int loss_kind;      // -1 none, 0 BCE, 1 squared error
    float* hsave;                  // optional [B][64][H][W]: the hidden map, kept for the backward pass (which then skips GEMM1)
};

#define DF_FWD_LDS_FLOATS (DF_LAT * 3 * DF_ZS + 2 * 32 * 64 + 4)

__device__ __forceinline__ int slot3(int r) { return (r + 3) % 3; }

__global__ void __launch_bounds__(256, 2) decoder_fused_fwd_kernel(const DecFwdArgs a) {
    extern __shared__ float lds[];
    float* zrow = lds;                          // [ci][slot 3][DF_ZS]
    float* pbuf = zrow + DF_LAT * 3 * DF_ZS;    // [mt][ct 32][64]
    float* red = pbuf + 2 * 32 * 64;            // [4]

    const DecGeom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mt = wv & 1, gp = wv >> 1, n = lane & 31, hh = lane >> 5;
    const int wi = blockIdx.x;
    const int band = wi % g.nbands, xt = (wi / g.nbands) % g.nxt, b = wi / (g.nbands * g.nxt);
    const int y0 = band * g.TH, y1 = min(g.H, y0 + g.TH);
    const int c0 = df_c0(xt, g.nxt, g.W);
    int own_lo, own_hi;
    df_own(xt, g.nxt, g.W, &own_lo, &own_hi);
    const size_t plane = (size_t)g.H * g.W;
    const float* zb = a.z + (size_t)b * DF_LAT * plane;

    // ---- weights: the A operands of both products are the same for every row a wave processes, so they live
    // in registers for the whole kernel (72 + 16 VGPRs); only the pixel operands go through LDS
    float wa[72], w2a[16], b1r[16];
#pragma unroll
    for (int s = 0; s < 72; ++s) wa[s] = a.pack[DF_OFF_W1P + s * 128 + mt * 64 + lane];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        w2a[j] = a.pack[DF_OFF_W2P + (mt * 16 + j) * 64 + lane];
        b1r[j] = a.pack[DF_OFF_B1 + mt * 32 + acc_row(j, hh)];
    }
    // ---- z row staging: thread t handles elements e = t + 256 i of the 16 x 66 row image
    int z_ch[DF_ZL], z_j[DF_ZL];
#pragma unroll
    for (int i = 0; i < DF_ZL; ++i) {
        int e = tid + 256 * i;
        if (e >= DF_NZ) e = DF_NZ - 1;          // duplicates of the last element: same value, harmless
        z_ch[i] = e / 66;
        z_j[i] = e - z_ch[i] * 66;
    }
    auto z_load = [&](int rr, float* v) {
        const bool rok = rr >= 0 && rr < g.H;
        const int rc = rok ? rr : 0;
#pragma unroll
        for (int i = 0; i < DF_ZL; ++i) {
            const int col = c0 - 1 + z_j[i];
            const bool ok = rok && col >= 0 && col < g.W;
            const float x = zb[(size_t)z_ch[i] * plane + (size_t)rc * g.W + (ok ? col : 0)];
            v[i] = ok ? x : 0.f;
        }
    };
    auto z_store = [&](int rr, const float* v) {
        const int s = slot3(rr);
#pragma unroll
        for (int i = 0; i < DF_ZL; ++i) zrow[(z_ch[i] * 3 + s) * DF_ZS + z_j[i]] = v[i];
    };
    {
        float v0[DF_ZL], v1[DF_ZL], v2[DF_ZL];      // all three rows in flight at once
        z_load(y0 - 2, v0);
        z_load(y0 - 1, v1);
        z_load(y0, v2);
        z_store(y0 - 2, v0);
        z_store(y0 - 1, v1);
        z_store(y0, v2);
    }
    __syncthreads();

    // ---- per-thread logit row accumulators: thread (c, x), three rolling rows
    const int oc = tid >> 6, ox = tid & 63;
    const bool outp = oc < g.C;
    const int ocol = c0 + ox;
    const bool owned = outp && ocol >= own_lo && ocol < own_hi;
    float la0 = 0.f, la1 = 0.f, la2 = 0.f, lsum = 0.f;
    const float b2c = a.pack[DF_OFF_B2 + (outp ? oc : 0)];

    for (int r = y0 - 1; r <= y1; ++r) {
        const bool row_in = r >= 0 && r < g.H;          // h row r exists (else it is conv2's zero padding)
        float zv[DF_ZL];
        z_load(r + 2, zv);
        const int yo = r - 1;                           // the output row completed by this iteration
        const bool fin = yo >= y0 && yo < y1;
        float tgt = 0.f;
        if (a.target && fin && owned) tgt = a.target[((size_t)b * g.C + oc) * plane + (size_t)yo * g.W + ocol];

        f32x16 d2 = {0};
        if (row_in) {
            f32x16 acc = {0};
            const float* zb0 = zrow + hh * 3 * DF_ZS + gp * 32 + n;
            const int s0 = slot3(r - 1) * DF_ZS, s1 = slot3(r) * DF_ZS, s2 = slot3(r + 1) * DF_ZS;
            // pixel operands of channel pair cp+1 are fetched from LDS while the 9 MFMAs of pair cp issue
            float bq[9], bn[9];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bq[kx] = zb0[s0 + kx];
                bq[3 + kx] = zb0[s1 + kx];
                bq[6 + kx] = zb0[s2 + kx];
            }
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) {
                if (cp < 7) {
                    const float* zc = zb0 + (cp + 1) * 2 * 3 * DF_ZS;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bn[kx] = zc[s0 + kx];
                        bn[3 + kx] = zc[s1 + kx];
                        bn[6 + kx] = zc[s2 + kx];
                    }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cp * 9 + t], bq[t], acc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) bq[t] = bn[t];
            }
            // bias + leaky_relu in registers; columns beyond the image are conv2's zero padding
            const bool colok = c0 + gp * 32 + n < g.W;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = acc[j] + b1r[j];
                v = v > 0.f ? v : 0.01f * v;
                acc[j] = colok ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[j], acc[j], d2, 0, 0, 0);
            // the hidden row for the backward pass: every (row, column) is written by the band / column tile that owns it
            // (register j of a lane half = one channel, its 32 lanes = 128 contiguous bytes)
            if (a.hsave && r >= y0 && r < y1) {
                const int hcol = c0 + gp * 32 + n;
                if (hcol >= own_lo && hcol < own_hi) {
                    float* hp = a.hsave + ((size_t)b * DF_HID + mt * 32 + 4 * hh) * plane + (size_t)r * g.W + hcol;
#pragma unroll
                    for (int j = 0; j < 16; ++j) hp[(size_t)((j & 3) + 8 * (j >> 2)) * plane] = acc[j];
                }
            }
        }
        __syncthreads();                                // z slots and pbuf of the previous row are free
        if (row_in) {
#pragma unroll
            for (int j = 0; j < 16; ++j) pbuf[(mt * 32 + acc_row(j, hh)) * 64 + gp * 32 + n] = d2[j];
        }
        z_store(r + 2, zv);
        __syncthreads();
        if (outp) {
            // la0 / la1 / la2 = the logit rows r-1 / r / r+1 under construction (rotated at the end)
            if (row_in) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = ox + kx - 1;
                    if (xx >= 0 && xx < 64) {
                        const float* pp = pbuf + (oc * 9 + kx) * 64 + xx;
                        s0 += pp[0] + pp[32 * 64];                        // ky = 0 -> row r+1
                        s1 += pp[3 * 64] + pp[(32 + 3) * 64];             // ky = 1 -> row r
                        s2 += pp[6 * 64] + pp[(32 + 6) * 64];             // ky = 2 -> row r-1
                    }
                }
                la2 += s0;
                la1 += s1;
                la0 += s2;
            }
            if (fin && owned) {
                const float l = la0 + b2c;
                if (a.logits) a.logits[((size_t)b * g.C + oc) * plane + (size_t)yo * g.W + ocol] = l;
                if (a.loss_kind >= 0) {
                    const float p = 1.f / (1.f + expf(-l));
                    if (a.loss_kind == 0) {
                        const float l1 = fmaxf(logf(p), -100.f), l0 = fmaxf(log1pf(-p), -100.f);
                        lsum -= tgt * l1 + (1.f - tgt) * l0;
                    } else {
                        const float d = tgt - p;
                        lsum = fmaf(d, d, lsum);
                    }
                }
            }
            la0 = la1;
            la1 = la2;
            la2 = 0.f;
        }
    }
    if (a.loss_kind >= 0) {
        const float s = wave_sum(lsum);
        if (lane == 0) red[wv] = s;
        __syncthreads();
        if (tid == 0) a.loss_part[wi] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// per_sample[b] = (sum of the sample's workgroup partials, in order) / n
__global__ void decoder_fused_loss_reduce_kernel(const float* __restrict__ part, int per_sample_parts, float inv_n, int B,
                                                 float* __restrict__ per_sample) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < per_sample_parts; ++i) s += part[(size_t)b * per_sample_parts + i];
    per_sample[b] = s * inv_n;
}

extern "C" int scm_decoder_fused_workspace(int B, int H, int W, int C, size_t* fwd_floats, size_t* bwd_floats) {
    SCM_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 1 && C <= 3, "scm_decoder_fused_workspace: bad shape");
    DecGeom g;
    df_geom(B, H, W, C, &g);
    if (fwd_floats) *fwd_floats = (size_t)B * g.nxt * g.nbands;
    if (bwd_floats) {
        DecGeom gb;
        df_geom_bwd(B, H, W, C, &gb);
        *bwd_floats = (size_t)B * gb.nxt * gb.nbands * DF_BWD_PART;
    }
    return SCM_OK;
}

extern "C" int scm_decoder_fused_fwd(const scm_decoder_fused_desc* d, void* stream) {
    SCM_REQUIRE(d != nullptr && d->z && d->pack, "scm_decoder_fused_fwd: z and pack required");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 1 && d->C <= 3, "scm_decoder_fused_fwd: bad shape B=%d H=%d W=%d C=%d",
                d->B, d->H, d->W, d->C);
    SCM_REQUIRE(d->loss_kind >= -1 && d->loss_kind <= 1, "scm_decoder_fused_fwd: loss_kind must be -1, 0 or 1");
    SCM_REQUIRE(d->loss_kind < 0 || (d->target && d->loss && d->workspace), "scm_decoder_fused_fwd: a loss needs target, loss and workspace");
    SCM_REQUIRE(d->logits || d->loss_kind >= 0, "scm_decoder_fused_fwd: nothing to compute");
    DecFwdArgs a;
    a.z = d->z; a.pack = d->pack; a.target = d->loss_kind >= 0 ? d->target : nullptr; a.logits = d->logits;
    a.loss_part = d->workspace; a.loss_kind = d->loss_kind; a.hsave = d->h_save;
    df_geom(d->B, d->H, d->W, d->C, &a.g);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = DF_FWD_LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(decoder_fused_fwd_kernel), attr_done, (int)lds)) return rc;
    const int wgs = d->B * a.g.nxt * a.g.nbands;
    hipLaunchKernelGGL(decoder_fused_fwd_kernel, dim3(wgs), dim3(256), lds, s, a);
    SCM_LAUNCH_CHECK("decoder_fused_fwd_kernel");
    if (d->loss_kind >= 0) {
        const float inv_n = 1.f / ((float)d->C * d->H * d->W);
        hipLaunchKernelGGL(decoder_fused_loss_reduce_kernel, dim3((d->B + 255) / 256), dim3(256), 0, s, d->workspace,
                           a.g.nxt * a.g.nbands, inv_n, d->B, d->loss);
        SCM_LAUNCH_CHECK("decoder_fused_loss_reduce_kernel");
    }
    return SCM_OK;
}

// ---------------------------------------------------------------------------------------- backward
// One workgroup (4 waves, one per SIMD, the whole register file) per band of rows x 64 columns, row by row.
// Nothing 64 channels wide touches HBM: h is recomputed from z (GEMM1 again), and per h row r
//   dl   = d loss / d logit from the saved logits and the target rows (staged in LDS, true and ownership-masked);
//   dh   = lrelu'(h) . sum_{c,tap} W2c[c][ch][tap] dl[c](r-ky+1, x-kx+1)      MFMA, K = 27: B gathered from the dl rows
//   Q    = W1c^T dh in scatter form, Q[(ci,tap)][px] = sum_co W1c[co][ci][tap] dh[co][px]: dh's accumulator registers are
//          the B operand; each lane ends up with all 9 taps of 8 input channels of its pixel, folds them over ky into
//          three rolling dz rows in registers, and a finished row goes through LDS once for the +-1 column shift;
//   dW1 += dh_owned x z(taps), dW2 += h x dl_owned(taps): contractions over PIXELS, so the accumulator tiles (pixel on
//          the lane) are transposed through LDS ([channel][pixel] images) and read back as A operands;
//          db1 rides along as a column of ones in the z gather, db2 is summed while dl is staged.
// Each workgroup writes one partial slab; decoder_fused_wgrad_reduce_kernel sums the slabs in a fixed order into the
// ConvTranspose2d layouts (taps flipped back, conv2's gradient broadcast over the 16 factors).
struct DecBwdArgs {
    const float* z; const float* pack; const float* logits; const float* target; const float* gscale;
    float* dz; float* part;
    DecGeom g; int loss_kind; float inv_n;
    const float* hsaved;           // HS = 1: the hidden map the forward pass kept ([B][64][H][W]); GEMM1 is not repeated
};

#define DB_ZS 72
#define DB_OFF_W1Q 0                                   // 10240
#define DB_OFF_Z (DB_OFF_W1Q + 10240)                  // [16][4][DB_ZS]
#define DB_OFF_DLT (DB_OFF_Z + DF_LAT * 4 * DB_ZS)     // [3][4][DB_ZS] true dl
#define DB_OFF_DLO (DB_OFF_DLT + 3 * 4 * DB_ZS)        // [3][4][DB_ZS] dl masked to the owned pixels
#define DB_OFF_HT (DB_OFF_DLO + 3 * 4 * DB_ZS)         // [64][65]
#define DB_OFF_DHT (DB_OFF_HT + 64 * 65)               // [64][65]
#define DB_OFF_VZ (DB_OFF_DHT + 64 * 65)               // [2][16][3][64]
#define DB_OFF_ONES (DB_OFF_VZ + 2 * 16 * 3 * 64)      // [72] ones, [72] zeros
#define DB_OFF_RED (DB_OFF_ONES + 144)                 // [16]
#define DB_LDS_FLOATS (DB_OFF_RED + 16)

__device__ __forceinline__ int slot4(int r) { return (r + 4) & 3; }

template <int HS>
__global__ void __launch_bounds__(256, 1) decoder_fused_bwd_kernel(const DecBwdArgs a) {
    extern __shared__ float lds[];
    float* w1q = lds + DB_OFF_W1Q;
    float* zrow = lds + DB_OFF_Z;
    float* dlt = lds + DB_OFF_DLT;
    float* dlo = lds + DB_OFF_DLO;
    float* hT = lds + DB_OFF_HT;
    float* dhT = lds + DB_OFF_DHT;
    float* vz = lds + DB_OFF_VZ;

    const DecGeom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mt = wv & 1, gp = wv >> 1, n = lane & 31, hh = lane >> 5;
    const int wi = blockIdx.x;
    const int band = wi % g.nbands, xt = (wi / g.nbands) % g.nxt, b = wi / (g.nbands * g.nxt);
    const int y0 = band * g.TH, y1 = min(g.H, y0 + g.TH);
    const int c0 = df_c0(xt, g.nxt, g.W);
    int own_lo, own_hi;
    df_own(xt, g.nxt, g.W, &own_lo, &own_hi);
    const size_t plane = (size_t)g.H * g.W;
    const float* zb = a.z + (size_t)b * DF_LAT * plane;
    const int C9 = 9 * g.C;

    // ---- resident operands
    float wa[HS ? 1 : 72], w2d[14], b1r[HS ? 1 : 16];
    if constexpr (!HS) {
#pragma unroll
        for (int s = 0; s < 72; ++s) wa[s] = a.pack[DF_OFF_W1P + s * 128 + mt * 64 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) b1r[j] = a.pack[DF_OFF_B1 + mt * 32 + acc_row(j, hh)];
    }
#pragma unroll
    for (int s = 0; s < 14; ++s) w2d[s] = a.pack[DF_OFF_W2D + (mt * 14 + s) * 64 + lane];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.pack + DF_OFF_W1Q);
        f32x4* dst = reinterpret_cast<f32x4*>(w1q);
        for (int i = tid; i < 10240 / 4; i += 256) dst[i] = src[i];
        if (tid < 144) lds[DB_OFF_ONES + tid] = tid < 72 ? 1.f : 0.f;
    }
    const float gs = a.gscale[b] * a.inv_n;

    // ---- staging maps: z rows (16 x 66), dl rows (C x 66)
    int z_ch[DF_ZL], z_j[DF_ZL];
#pragma unroll
    for (int i = 0; i < DF_ZL; ++i) {
        int e = tid + 256 * i;
        if (e >= DF_NZ) e = DF_NZ - 1;
        z_ch[i] = e / 66;
        z_j[i] = e - z_ch[i] * 66;
    }
    auto z_load = [&](int rr, float* v) {
        const bool rok = rr >= 0 && rr < g.H;
        const int rc = rok ? rr : 0;
#pragma unroll
        for (int i = 0; i < DF_ZL; ++i) {
            const int col = c0 - 1 + z_j[i];
            const bool ok = rok && col >= 0 && col < g.W;
            const float x = zb[(size_t)z_ch[i] * plane + (size_t)rc * g.W + (ok ? col : 0)];
            v[i] = ok ? x : 0.f;
        }
    };
    auto z_store = [&](int rr, const float* v) {
        const int s = slot4(rr);
#pragma unroll
        for (int i = 0; i < DF_ZL; ++i) zrow[(z_ch[i] * 4 + s) * DB_ZS + z_j[i]] = v[i];
    };
    const bool dl_thr = tid < g.C * 66;
    const int dl_c = dl_thr ? tid / 66 : 0, dl_j = dl_thr ? tid - dl_c * 66 : 0;
    const int dl_col = c0 - 1 + dl_j;
    const bool dl_colok = dl_thr && dl_col >= 0 && dl_col < g.W;
    const bool dl_colown = dl_colok && dl_col >= own_lo && dl_col < own_hi;
    float db2 = 0.f;
    // (logit, target) of row rr at this thread's (c, column), and the gradient from them
    auto dl_load = [&](int rr, float* lg, float* tg) {
        const bool ok = dl_colok && rr >= 0 && rr < g.H;
        const size_t o = ((size_t)b * g.C + dl_c) * plane + (size_t)(ok ? rr : 0) * g.W + (ok ? dl_col : 0);
        *lg = a.logits[o];
        *tg = a.target[o];
    };
    auto dl_store = [&](int rr, float lg, float tg) {
        if (!dl_thr) return;
        const bool ok = dl_colok && rr >= 0 && rr < g.H;
        const float p = 1.f / (1.f + expf(-lg));
        const float pq = p * (1.f - p);
        // BCE: torch's binary_cross_entropy_backward (p - t) / max(p (1-p), 1e-12), times dp/dlogit = p (1-p)
        const float dp = a.loss_kind == 0 ? (p - tg) / fmaxf(pq, 1e-12f) : 2.f * (p - tg);
        const float v = ok ? gs * dp * pq : 0.f;
        const float vo = (dl_colown && rr >= y0 && rr < y1) ? v : 0.f;
        const int o = (dl_c * 4 + slot4(rr)) * DB_ZS + dl_j;
        dlt[o] = v;
        dlo[o] = vo;
        db2 += vo;
    };
    {
        float v0[DF_ZL], v1[DF_ZL], v2[DF_ZL], l0, l1, l2, t0, t1, t2;
        z_load(y0 - 2, v0); z_load(y0 - 1, v1); z_load(y0, v2);
        dl_load(y0 - 2, &l0, &t0); dl_load(y0 - 1, &l1, &t1); dl_load(y0, &l2, &t2);
        z_store(y0 - 2, v0); z_store(y0 - 1, v1); z_store(y0, v2);
        dl_store(y0 - 2, l0, t0); dl_store(y0 - 1, l1, t1); dl_store(y0, l2, t2);
    }
    __syncthreads();

    // ---- per-lane gather maps
    // dW1's B operand: column q = nt*32 + n of [ (ci,tap) 0..143 | ones | zeros ]
    int w1b_off[5], w1b_ky[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int q = nt * 32 + n;
        if (q < 144) {
            const int ci = q / 9, tap = q - ci * 9;
            w1b_off[nt] = DB_OFF_Z + ci * 4 * DB_ZS + tap % 3;
            w1b_ky[nt] = tap / 3;
        } else {
            w1b_off[nt] = q == 144 ? DB_OFF_ONES : DB_OFF_ONES + 72;
            w1b_ky[nt] = 3;
        }
    }
    // dW2's B operand: column ct = n of [ (c,tap) | zeros ]
    int w2b_off, w2b_ky;
    if (n < C9) {
        const int c = n / 9, tap = n - c * 9;
        w2b_off = DB_OFF_DLO + c * 4 * DB_ZS + 2 - tap % 3;
        w2b_ky = tap / 3;
    } else {
        w2b_off = DB_OFF_ONES + 72;
        w2b_ky = 3;
    }
    const int kh = gp;                      // pixel half of the K range in the weight-gradient products
    const int px0 = kh * 32 + hh;

    f32x16 dW1acc[5], dW2acc = {0};
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) dW1acc[nt] = (f32x16){0};
    float V[3][8][3];                       // dz rows r-1 / r / r+1 under construction: [row][ci local][kx]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) V[i][c][k] = 0.f;

    const int ox = tid & 63, ocq = tid >> 6;
    const int ocol = c0 + ox;
    const bool ocolown = ocol >= own_lo && ocol < own_hi;

    // HS: this lane's 16 channels of hidden row rr at its pixel, requested one row ahead (zero outside the image, as
    // the forward pass wrote conv2's zero padding)
    float hcur[HS ? 16 : 1], hnext[HS ? 16 : 1];
    auto h_load = [&](int rr, float* v) {
        if constexpr (HS) {
            const int hcol = c0 + gp * 32 + n;
            const bool ok = rr >= 0 && rr < g.H && hcol < g.W;
            const float* hp = a.hsaved + ((size_t)b * DF_HID + mt * 32 + 4 * hh) * plane + (size_t)(ok ? rr : 0) * g.W +
                              (ok ? hcol : 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float x = hp[(size_t)((j & 3) + 8 * (j >> 2)) * plane];
                v[j] = ok ? x : 0.f;
            }
        }
    };
    h_load(y0 - 1, hcur);

    for (int r = y0 - 1; r <= y1; ++r) {
        const bool row_in = r >= 0 && r < g.H;
        const bool row_own = r >= y0 && r < y1;
        float zv[DF_ZL], lgv, tgv;
        z_load(r + 2, zv);
        dl_load(r + 2, &lgv, &tgv);
        h_load(r + 1, hnext);
        // LDS row offsets of image rows r-1, r, r+1 (and 0 for the constant regions)
        const int ro0 = slot4(r - 1) * DB_ZS, ro1 = slot4(r) * DB_ZS, ro2 = slot4(r + 1) * DB_ZS;

        f32x16 hacc = {0}, dh = {0};
        if (row_in) {
            if constexpr (!HS) {
            // ---- GEMM1: h row r
            const float* zb0 = zrow + hh * 4 * DB_ZS + gp * 32 + n;
            float bq[9], bn[9];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bq[kx] = zb0[ro0 + kx];
                bq[3 + kx] = zb0[ro1 + kx];
                bq[6 + kx] = zb0[ro2 + kx];
            }
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) {
                if (cp < 7) {
                    const float* zc = zb0 + (cp + 1) * 2 * 4 * DB_ZS;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bn[kx] = zc[ro0 + kx];
                        bn[3 + kx] = zc[ro1 + kx];
                        bn[6 + kx] = zc[ro2 + kx];
                    }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cp * 9 + t], bq[t], hacc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) bq[t] = bn[t];
            }
            }
            // ---- dh = lrelu'(h) . (W2c^T dl): K = (c,tap) pairs 2s + hh
            float bd[14];
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                const int ct = 2 * s + hh;
                const int c = ct / 9, tap = ct - c * 9, ky = tap / 3, kx = tap - ky * 3;
                const int ro = ky == 0 ? ro2 : (ky == 1 ? ro1 : ro0);           // dl row r - ky + 1
                const int off = ct < C9 ? DB_OFF_DLT + c * 4 * DB_ZS + ro + 2 - kx + gp * 32 + n : DB_OFF_ONES + 72;
                bd[s] = lds[off];
            }
#pragma unroll
            for (int s = 0; s < 14; ++s) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(w2d[s], bd[s], dh, 0, 0, 0);
            const int hcol = c0 + gp * 32 + n;
            const bool colok = hcol < g.W;
            const bool hown = row_own && hcol >= own_lo && hcol < own_hi;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float hv, dv;
                if constexpr (HS) {
                    // (leaky_relu keeps the sign: the saved activation is its own gate; zero where the forward masked it)
                    hv = hcur[j];
                    dv = colok ? dh[j] * (hv > 0.f ? 1.f : 0.01f) : 0.f;
                } else {
                    const float pre = hacc[j] + b1r[j];
                    hv = colok ? (pre > 0.f ? pre : 0.01f * pre) : 0.f;
                    dv = colok ? dh[j] * (pre > 0.f ? 1.f : 0.01f) : 0.f;
                }
                hacc[j] = hv;
                dh[j] = dv;
                // [channel][pixel] images for the products that contract over pixels
                const int o = (mt * 32 + acc_row(j, hh)) * 65 + gp * 32 + n;
                hT[o] = hv;
                dhT[o] = hown ? dv : 0.f;
            }
        }
        if constexpr (HS) {
#pragma unroll
            for (int j = 0; j < 16; ++j) hcur[j] = hnext[j];
        }
        z_store(r + 2, zv);
        dl_store(r + 2, lgv, tgv);
        __syncthreads();                    // images of row r, staged rows r+2 visible

        if (row_in) {
            // ---- weight gradients: K = this wave's 32 pixels, two per step
            {
                const float* ap = dhT + (mt * 32 + n) * 65 + px0;
                int bo[5];
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
                    bo[nt] = w1b_off[nt] + (w1b_ky[nt] == 0 ? ro0 : (w1b_ky[nt] == 1 ? ro1 : (w1b_ky[nt] == 2 ? ro2 : 0))) + px0;
                float av = ap[0], bv[5], an = 0.f, bnx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) bv[nt] = lds[bo[nt]];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s < 15) {
                        an = ap[2 * s + 2];
#pragma unroll
                        for (int nt = 0; nt < 5; ++nt) bnx[nt] = lds[bo[nt] + 2 * s + 2];
                    }
                    if (row_own) {
#pragma unroll
                        for (int nt = 0; nt < 5; ++nt) dW1acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nt], dW1acc[nt], 0, 0, 0);
                    }
                    av = an;
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) bv[nt] = bnx[nt];
                }
            }
            {
                // dW2[ch][(c,ky,kx)] += h[ch](r, px) dl_owned[c](r - ky + 1, px - kx + 1)
                const float* ap = hT + (mt * 32 + n) * 65 + px0;
                const int bo = w2b_off + (w2b_ky == 0 ? ro2 : (w2b_ky == 1 ? ro1 : (w2b_ky == 2 ? ro0 : 0))) + px0;
                float av[16], bv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    av[s] = ap[2 * s];
                    bv[s] = lds[bo + 2 * s];
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) dW2acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], dW2acc, 0, 0, 0);
            }
            // ---- Q = W1c^T dh (this wave's 32 output channels of conv1), folded over ky into the rolling rows
            {
                f32x16 Q[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) Q[t] = (f32x16){0};
                const float* qa = w1q + mt * 5120 + lane;
                float av[5], an[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 5; ++t) av[t] = qa[t * 1024];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < 15) {
#pragma unroll
                        for (int t = 0; t < 5; ++t) an[t] = qa[t * 1024 + (j + 1) * 64];
                    }
#pragma unroll
                    for (int t = 0; t < 5; ++t) Q[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], dh[j], Q[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 5; ++t) av[t] = an[t];
                }
                // dz[ci](r + ky - 1, x' + kx - 1) += W1c[co][ci][ky][kx] dh[co](r, x')
#pragma unroll
                for (int jj = 0; jj < 72; ++jj) {
                    const int cl = jj / 9, tap = jj % 9, ky = tap / 3, kx = tap % 3;
                    V[ky][cl][kx] += Q[jj / 16][jj % 16];
                }
            }
        }
        // ---- dz row r-1 is complete: partial (over this wave's channel half) columns go through LDS for the shift
        const int yo = r - 1;
        const bool fin = yo >= y0 && yo < y1;
        if (fin) {
#pragma unroll
            for (int cl = 0; cl < 8; ++cl)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) vz[((mt * 16 + 8 * hh + cl) * 3 + kx) * 64 + gp * 32 + n] = V[0][cl][kx];
        }
#pragma unroll
        for (int cl = 0; cl < 8; ++cl)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                V[0][cl][kx] = V[1][cl][kx];
                V[1][cl][kx] = V[2][cl][kx];
                V[2][cl][kx] = 0.f;
            }
        __syncthreads();                    // vz complete; hT / dhT reads of this row done
        if (fin) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ci = ocq * 4 + i;
                float s = 0.f;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xs = ox + 1 - kx;
                    if (xs >= 0 && xs < 64) s += vz[(ci * 3 + kx) * 64 + xs] + vz[((16 + ci) * 3 + kx) * 64 + xs];
                }
                if (ocolown) a.dz[((size_t)b * DF_LAT + ci) * plane + (size_t)yo * g.W + ocol] = s;
            }
        }
        // (the next iteration's vz / image writes come after its own first barrier)
    }

    // ---- partial slab of this workgroup
    float* part = a.part + (size_t)wi * DF_BWD_PART;
    // dW1: wave (mt, kh) holds rows co = mt*32 + .. over its pixel half: the two halves add through LDS (fixed order)
    float* xb = hT;                         // 64 x 65 scratch, reused
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        if (kh == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) xb[(mt * 32 + acc_row(j, hh)) * 65 + n] = dW1acc[nt][j];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = mt * 32 + acc_row(j, hh);
                part[co * 160 + nt * 32 + n] = dW1acc[nt][j] + xb[co * 65 + n];
            }
        }
        __syncthreads();
    }
    if (kh == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xb[(mt * 32 + acc_row(j, hh)) * 65 + n] = dW2acc[j];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ch = mt * 32 + acc_row(j, hh);
            part[64 * 160 + ch * 32 + n] = dW2acc[j] + xb[ch * 65 + n];
        }
    }
    // db2[c]: threads (c, j) hold their column sums; 66 per channel, summed in lane order by one thread
    __syncthreads();
    float* cs = dhT;                        // [C][66]
    if (dl_thr) cs[dl_c * 66 + dl_j] = db2;
    __syncthreads();
    if (tid < 4) {
        float s = 0.f;
        if (tid < g.C) for (int j = 0; j < 66; ++j) s += cs[tid * 66 + j];
        part[64 * 160 + 64 * 32 + tid] = s;
    }
}

// Sum the workgroup slabs (fixed order) into the ConvTranspose2d gradient layouts:
//   dWt1[ci][co][2-ky][2-kx] = sum dW1[co][ci*9 + ky*3 + kx],  db1[co] = sum dW1[co][144]
//   dWt2[ch][f*C + c][2-ky][2-kx] = sum dW2[ch][c*9 + ky*3 + kx] for every factor f,  db2[f*C + c] = sum db2[c]
// accumulate: dWt1/db1/dWt2/db2 += (grad sinks), else overwrite.
__global__ void __launch_bounds__(256) decoder_fused_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int L, int C,
                                                                         int accumulate, float* __restrict__ dwt1,
                                                                         float* __restrict__ db1, float* __restrict__ dwt2,
                                                                         float* __restrict__ db2) {
    // 32 outputs x 8 slab ranges per workgroup: range q sums its slabs in order, the 8 range sums add in order
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int n1 = 64 * 144, nb1 = 64, n2 = 64 * 9 * C, nb2 = C;
    const bool live = i < n1 + nb1 + n2 + nb2;
    int src = 0;
    if (i < n1) { const int co = i / 144, k = i - co * 144; src = co * 160 + k; }
    else if (i < n1 + nb1) src = (i - n1) * 160 + 144;
    else if (i < n1 + nb1 + n2) { const int k = i - n1 - nb1, ch = k / (9 * C), ct = k - ch * 9 * C; src = 64 * 160 + ch * 32 + ct; }
    else if (live) src = 64 * 160 + 64 * 32 + (i - n1 - nb1 - n2);
    const int per = (nparts + 7) / 8, p0 = q * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    if (live) for (int p = p0; p < p1; ++p) s += part[(size_t)p * DF_BWD_PART + src];
    red[q][e] = s;
    __syncthreads();
    if (q != 0 || !live) return;
    s = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
    if (i < n1) {
        const int co = i / 144, k = i - co * 144, ci = k / 9, tap = k - ci * 9, ky = tap / 3, kx = tap - ky * 3;
        float* d = dwt1 + ((ci * DF_HID + co) * 3 + (2 - ky)) * 3 + (2 - kx);
        *d = accumulate ? *d + s : s;
    } else if (i < n1 + nb1) {
        float* d = db1 + (i - n1);
        *d = accumulate ? *d + s : s;
    } else if (i < n1 + nb1 + n2) {
        const int k = i - n1 - nb1, ch = k / (9 * C), ct = k - ch * 9 * C, c = ct / 9, tap = ct - c * 9, ky = tap / 3, kx = tap - ky * 3;
        for (int f = 0; f < L; ++f) {
            float* d = dwt2 + ((ch * (L * C) + f * C + c) * 3 + (2 - ky)) * 3 + (2 - kx);
            *d = accumulate ? *d + s : s;
        }
    } else {
        const int c = i - n1 - nb1 - n2;
        for (int f = 0; f < L; ++f) {
            float* d = db2 + f * C + c;
            *d = accumulate ? *d + s : s;
        }
    }
}

extern "C" int scm_decoder_fused_bwd(const scm_decoder_fused_bwd_desc* d, void* stream) {
    SCM_REQUIRE(d != nullptr && d->z && d->pack && d->logits && d->target && d->g_loss, "scm_decoder_fused_bwd: null input");
    SCM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 1 && d->C <= 3, "scm_decoder_fused_bwd: bad shape B=%d H=%d W=%d C=%d",
                d->B, d->H, d->W, d->C);
    SCM_REQUIRE(d->loss_kind == 0 || d->loss_kind == 1, "scm_decoder_fused_bwd: loss_kind must be 0 (BCE) or 1 (squared error)");
    SCM_REQUIRE(d->dz && d->workspace && d->d_conv1_weight && d->d_conv1_bias && d->d_conv2_weight && d->d_conv2_bias,
                "scm_decoder_fused_bwd: null output");
    DecBwdArgs a;
    a.z = d->z; a.pack = d->pack; a.logits = d->logits; a.target = d->target; a.gscale = d->g_loss;
    a.dz = d->dz; a.part = d->workspace; a.loss_kind = d->loss_kind; a.hsaved = d->h_saved;
    a.inv_n = 1.f / ((float)d->C * d->H * d->W);
    df_geom_bwd(d->B, d->H, d->W, d->C, &a.g);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = DB_LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    static std::atomic<uint64_t> attr_done_hs{0};
    const int wgs = d->B * a.g.nxt * a.g.nbands;
    if (d->h_saved) {
        if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(decoder_fused_bwd_kernel<1>), attr_done_hs, (int)lds)) return rc;
        hipLaunchKernelGGL(decoder_fused_bwd_kernel<1>, dim3(wgs), dim3(256), lds, s, a);
    } else {
        if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(decoder_fused_bwd_kernel<0>), attr_done, (int)lds)) return rc;
        hipLaunchKernelGGL(decoder_fused_bwd_kernel<0>, dim3(wgs), dim3(256), lds, s, a);
    }
    SCM_LAUNCH_CHECK("decoder_fused_bwd_kernel");
    const int total = 64 * 144 + 64 + 64 * 9 * d->C + d->C;
    hipLaunchKernelGGL(decoder_fused_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, s, d->workspace, wgs,
                       DF_LAT, d->C, d->accumulate, d->d_conv1_weight, d->d_conv1_bias, d->d_conv2_weight, d->d_conv2_bias);
    SCM_LAUNCH_CHECK("decoder_fused_wgrad_reduce_kernel");
    return SCM_OK;
}
