// Fused reward-head convolutions: RewardPredictor.conv1 (16 -> 32, 3x3 valid) + leaky_relu + conv2 (32 -> 3R, 3x3,
// stride 2, valid) of models.py:235-240 in ONE pass over the latent z, forward and backward; the softmax tail
// (models.py:241-250) stays scm_reward_tail.  Same construction as decoder_fused.hip (which see), with a 32-channel
// hidden map (one 32x32 MFMA row tile, two waves per workgroup) that never leaves the registers:
//   h      = lrelu(b1 + conv3x3(z, W1)) evaluated in same-size coordinates; the valid conv is its interior, and
//            only interior positions are ever consumed;
//   P      = W2 h in scatter form, P[(q,ky,kx)][px] = sum_ch W2[q][ch][ky][kx] h[ch][px] (54 of 64 rows for R = 2),
//            the accumulator of the first product being the B operand of the second;
//   o[q](i, j) = b2[q] + sum_{ky,kx} P[(q,ky,kx)](2i + ky + 1, 2j + kx + 1): the full-resolution tap sum of the decoder
//            kernel sampled at the even positions (2i + 2, 2j + 2).
// Backward: do (gradient of o, from scm_reward_tail_bwd) is zero-inserted on the fly into full resolution
// (dl'(y, x) = do((y-2)/2, (x-2)/2) at even y, x), after which every product is the decoder's: dh = lrelu'(h) . W2^T dl',
// dz through Q = W1^T dh folded over taps in registers, dW1 / dW2 through [channel][pixel] LDS images, partial slabs
// per workgroup summed in a fixed order.  Frames up to 64 wide; wider ones keep the layer-by-layer path.
#include "common.h"

#define RF_LAT 16
#define RF_HID 32
#define RF_ZS 72
#define RF_NZ (RF_LAT * 66)
#define RF_NT 128
#define RF_ZL ((RF_NZ + RF_NT - 1) / RF_NT)

// pack layout (floats)
#define RF_OFF_W1P 0                         // [72 k-steps][64]            GEMM1 A operand
#define RF_OFF_W2P (72 * 64)                 // [2 tiles][16 regs][64]      P = W2 h A operand
#define RF_OFF_B1 (RF_OFF_W2P + 2 * 16 * 64) // [32]
#define RF_OFF_B2 (RF_OFF_B1 + 32)           // [8]
#define RF_OFF_W1Q (RF_OFF_B2 + 8)           // [5 tiles][16 regs][64]      Q = W1^T dh A operand
#define RF_OFF_W2D (RF_OFF_W1Q + 5 * 16 * 64)// [27 k-steps][64]            dh = W2^T dl' A operand
#define RF_PACK_FLOATS (RF_OFF_W2D + 27 * 64)

extern "C" size_t scm_reward_fused_pack_floats(void) { return RF_PACK_FLOATS; }

__device__ __forceinline__ int rf_acc_row(int j, int hh) { return (j & 3) + 8 * (j >> 2) + 4 * hh; }
__device__ __forceinline__ int rf_slot3(int r) { return (r + 3) % 3; }
__device__ __forceinline__ int rf_slot4(int r) { return (r + 4) & 3; }

// conv1.weight (32, 16, 3, 3), conv2.weight (3R, 32, 3, 3): nn.Conv2d layouts
__global__ void reward_fused_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                         const float* __restrict__ w2, const float* __restrict__ b2, int Q,
                                         float* __restrict__ pack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= RF_PACK_FLOATS) return;
    auto W1 = [&](int co, int ci, int tap) { return w1[(co * RF_LAT + ci) * 9 + tap]; };
    auto W2 = [&](int q, int ch, int tap) { return w2[(q * RF_HID + ch) * 9 + tap]; };
    float v = 0.f;
    if (i < RF_OFF_W2P) {                                           // w1p[s][l]
        const int l = i & 63, s = i >> 6, cp = s / 9, tap = s - cp * 9;
        v = W1(l & 31, 2 * cp + (l >> 5), tap);
    } else if (i < RF_OFF_B1) {                                     // w2p[t][j][l]: row ct = t*32 + (l&31), k = ch
        const int e = i - RF_OFF_W2P, l = e & 63, j = (e >> 6) & 15, t = e >> 10;
        const int ct = t * 32 + (l & 31);
        if (ct < 9 * Q) v = W2(ct / 9, rf_acc_row(j, l >> 5), ct % 9);
    } else if (i < RF_OFF_B2) {
        v = b1[i - RF_OFF_B1];
    } else if (i < RF_OFF_W1Q) {
        const int q = i - RF_OFF_B2;
        if (q < Q) v = b2[q];
    } else if (i < RF_OFF_W2D) {                                    // w1q[t][j][l], see decoder_fused.hip
        const int e = i - RF_OFF_W1Q, l = e & 63, j = (e >> 6) & 15, t = e >> 10;
        const int mm = l & 31, hm = (mm >> 2) & 1, reg = (mm >> 3) * 4 + (mm & 3), jj = t * 16 + reg;
        if (jj < 72) v = W1(rf_acc_row(j, l >> 5), 8 * hm + jj / 9, jj % 9);
    } else {                                                        // w2d[s][l]: row m = ch, k = ct = 2s + (l>>5)
        const int e = i - RF_OFF_W2D, l = e & 63, s = e >> 6;
        const int ct = 2 * s + (l >> 5);
        if (ct < 9 * Q) v = W2(ct / 9, l & 31, ct % 9);
    }
    pack[i] = v;
}

extern "C" int scm_reward_fused_pack(const float* conv1_weight, const float* conv1_bias, const float* conv2_weight,
                                     const float* conv2_bias, int latent, int Q, float* pack, void* stream) {
    SCM_REQUIRE(conv1_weight && conv1_bias && conv2_weight && conv2_bias && pack, "scm_reward_fused_pack: null pointer");
    SCM_REQUIRE(latent == RF_LAT && Q >= 1 && Q <= 6,
                "scm_reward_fused_pack: the fused reward head serves latent_dim 16 and up to 6 conv2 channels (got %d, %d)",
                latent, Q);
    hipLaunchKernelGGL(reward_fused_pack_kernel, dim3((RF_PACK_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       conv1_weight, conv1_bias, conv2_weight, conv2_bias, Q, pack);
    SCM_LAUNCH_CHECK("reward_fused_pack_kernel");
    return SCM_OK;
}

struct RewGeom {
    int B, H, W, Q, OH, OW, TH, nbands, ylo, yhi;     // bands partition the full-resolution rows [ylo, yhi)
};
static int rf_geom(int B, int H, int W, int Q, int min_wgs, RewGeom* g) {
    if (H < 5 || W < 5 || W > 64) return 0;
    g->B = B; g->H = H; g->W = W; g->Q = Q;
    g->OH = (H - 5) / 2 + 1;
    g->OW = (W - 5) / 2 + 1;
    int th = 64;
    const int span_f = 2 * g->OH - 1;                 // forward: rows 2 .. 2*OH (odd rows in between carry no output)
    (void)span_f;
    g->ylo = 0; g->yhi = H;
    while (th > 4 && (long)B * ((H + th - 1) / th) < min_wgs) th >>= 1;
    if (th > H) th = H;
    g->TH = th;
    g->nbands = (H + th - 1) / th;
    return 1;
}

// ----------------------------------------------------------------------------------------- forward
struct RewFwdArgs {
    const float* z; const float* pack; float* o; RewGeom g;
    float* hsave;       // optional [B][32][H][W]: the hidden map in same-size coordinates, kept for the backward pass
};
#define RF_FWD_LDS_FLOATS (RF_LAT * 3 * RF_ZS + 64 * 64)

__global__ void __launch_bounds__(RF_NT, 2) reward_fused_fwd_kernel(const RewFwdArgs a) {
    extern __shared__ float lds[];
    float* zrow = lds;                          // [ci][slot 3][RF_ZS]
    float* pbuf = zrow + RF_LAT * 3 * RF_ZS;    // [ct 64][64]
    const RewGeom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, gp = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int wi = blockIdx.x;
    const int band = wi % g.nbands, b = wi / g.nbands;
    const int y0 = band * g.TH, y1 = min(g.H, y0 + g.TH);
    const size_t plane = (size_t)g.H * g.W;
    const float* zb = a.z + (size_t)b * RF_LAT * plane;

    float wa[72], w2a[2][16], b1r[16];
#pragma unroll
    for (int s = 0; s < 72; ++s) wa[s] = a.pack[RF_OFF_W1P + s * 64 + lane];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        w2a[0][j] = a.pack[RF_OFF_W2P + j * 64 + lane];
        w2a[1][j] = a.pack[RF_OFF_W2P + (16 + j) * 64 + lane];
        b1r[j] = a.pack[RF_OFF_B1 + rf_acc_row(j, hh)];
    }
    int z_ch[RF_ZL], z_j[RF_ZL];
#pragma unroll
    for (int i = 0; i < RF_ZL; ++i) {
        int e = tid + RF_NT * i;
        if (e >= RF_NZ) e = RF_NZ - 1;
        z_ch[i] = e / 66;
        z_j[i] = e - z_ch[i] * 66;
    }
    auto z_load = [&](int rr, float* v) {
        const bool rok = rr >= 0 && rr < g.H;
        const int rc = rok ? rr : 0;
#pragma unroll
        for (int i = 0; i < RF_ZL; ++i) {
            const int col = z_j[i] - 1;
            const bool ok = rok && col >= 0 && col < g.W;
            const float x = zb[(size_t)z_ch[i] * plane + (size_t)rc * g.W + (ok ? col : 0)];
            v[i] = ok ? x : 0.f;
        }
    };
    auto z_store = [&](int rr, const float* v) {
        const int s = rf_slot3(rr);
#pragma unroll
        for (int i = 0; i < RF_ZL; ++i) zrow[(z_ch[i] * 3 + s) * RF_ZS + z_j[i]] = v[i];
    };
    {
        float v0[RF_ZL], v1[RF_ZL], v2[RF_ZL];
        z_load(y0 - 2, v0); z_load(y0 - 1, v1); z_load(y0, v2);
        z_store(y0 - 2, v0); z_store(y0 - 1, v1); z_store(y0, v2);
    }
    __syncthreads();

    // outputs (q, j): slot e = tid + 128 k (k < 2), q = e / 32, j = e % 32; full-resolution column x = 2j + 2
    int oq[2], oj[2];
    bool oact[2];
    float la[2][3], b2q[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tid + RF_NT * k;
        oq[k] = e >> 5;
        oj[k] = e & 31;
        oact[k] = oq[k] < g.Q && oj[k] < g.OW;
        b2q[k] = a.pack[RF_OFF_B2 + (oact[k] ? oq[k] : 0)];
        la[k][0] = la[k][1] = la[k][2] = 0.f;
    }

    for (int r = y0 - 1; r <= y1; ++r) {
        const bool row_in = r >= 0 && r < g.H;
        float zv[RF_ZL];
        z_load(r + 2, zv);
        const int yo = r - 1;
        const int oi = (yo - 2) >> 1;
        const bool fin = yo >= y0 && yo < y1 && yo >= 2 && (yo & 1) == 0 && oi < g.OH;
        f32x16 d2[2] = {{0}, {0}};
        if (row_in) {
            f32x16 acc = {0};
            const float* zb0 = zrow + hh * 3 * RF_ZS + gp * 32 + n;
            const int s0 = rf_slot3(r - 1) * RF_ZS, s1 = rf_slot3(r) * RF_ZS, s2 = rf_slot3(r + 1) * RF_ZS;
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
                    const float* zc = zb0 + (cp + 1) * 2 * 3 * RF_ZS;
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
            const bool colok = gp * 32 + n < g.W;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = acc[j] + b1r[j];
                v = v > 0.f ? v : 0.01f * v;
                acc[j] = colok ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                d2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[0][j], acc[j], d2[0], 0, 0, 0);
                d2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[1][j], acc[j], d2[1], 0, 0, 0);
            }
            // the hidden row for the backward pass (written by the band that owns the row)
            if (a.hsave && r >= y0 && r < y1 && colok) {
                float* hp = a.hsave + ((size_t)b * RF_HID + 4 * hh) * plane + (size_t)r * g.W + gp * 32 + n;
#pragma unroll
                for (int j = 0; j < 16; ++j) hp[(size_t)((j & 3) + 8 * (j >> 2)) * plane] = acc[j];
            }
        }
        __syncthreads();
        if (row_in) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) pbuf[(t * 32 + rf_acc_row(j, hh)) * 64 + gp * 32 + n] = d2[t][j];
        }
        z_store(r + 2, zv);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (oact[k]) {
                if (row_in) {
                    const float* pp = pbuf + (oq[k] * 9) * 64 + 2 * oj[k] + 1;     // column x + kx - 1 = 2j + 1 + kx
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        s0 += pp[kx * 64 + kx];
                        s1 += pp[(3 + kx) * 64 + kx];
                        s2 += pp[(6 + kx) * 64 + kx];
                    }
                    la[k][2] += s0;          // ky = 0 -> row r + 1
                    la[k][1] += s1;
                    la[k][0] += s2;
                }
                if (fin) a.o[(((size_t)b * g.Q + oq[k]) * g.OH + oi) * g.OW + oj[k]] = la[k][0] + b2q[k];
            }
            la[k][0] = la[k][1];
            la[k][1] = la[k][2];
            la[k][2] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------- backward
#define RF_BWD_PART (32 * 160 + 32 * 64 + 8)
struct RewBwdArgs {
    const float* z; const float* pack; const float* d_o; float* dz; float* part; RewGeom g;
    const float* hsaved;        // HS = 1: the forward's hidden map; conv1 is not repeated
};
#define RB_OFF_W1Q 0                                   // 5120
#define RB_OFF_Z (RB_OFF_W1Q + 5120)                   // [16][4][RF_ZS]
#define RB_OFF_DLT (RB_OFF_Z + RF_LAT * 4 * RF_ZS)     // [6][4][RF_ZS]
#define RB_OFF_DLO (RB_OFF_DLT + 6 * 4 * RF_ZS)        // [6][4][RF_ZS]
#define RB_OFF_HT (RB_OFF_DLO + 6 * 4 * RF_ZS)         // [32][65]
#define RB_OFF_DHT (RB_OFF_HT + 32 * 65)               // [32][65]
#define RB_OFF_VZ RB_OFF_HT                             // [16][3][64], aliases the two images (barriers in between)
#define RB_OFF_ONES (RB_OFF_DHT + 32 * 65)             // [72] ones, [72] zeros
#define RB_LDS_FLOATS (RB_OFF_ONES + 144)
#define RF_DL_N (6 * 66)
#define RF_DLL ((RF_DL_N + RF_NT - 1) / RF_NT)

template <int HS>
__global__ void __launch_bounds__(RF_NT, 1) reward_fused_bwd_kernel(const RewBwdArgs a) {
    extern __shared__ float lds[];
    float* w1q = lds + RB_OFF_W1Q;
    float* zrow = lds + RB_OFF_Z;
    float* dlt = lds + RB_OFF_DLT;
    float* dlo = lds + RB_OFF_DLO;
    float* hT = lds + RB_OFF_HT;
    float* dhT = lds + RB_OFF_DHT;
    float* vz = lds + RB_OFF_VZ;
    const RewGeom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, gp = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int wi = blockIdx.x;
    const int band = wi % g.nbands, b = wi / g.nbands;
    const int y0 = band * g.TH, y1 = min(g.H, y0 + g.TH);
    const size_t plane = (size_t)g.H * g.W;
    const float* zb = a.z + (size_t)b * RF_LAT * plane;
    const int C9 = 9 * g.Q;

    float wa[HS ? 1 : 72], w2d[27], b1r[HS ? 1 : 16];
    if constexpr (!HS) {
#pragma unroll
        for (int s = 0; s < 72; ++s) wa[s] = a.pack[RF_OFF_W1P + s * 64 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) b1r[j] = a.pack[RF_OFF_B1 + rf_acc_row(j, hh)];
    }
#pragma unroll
    for (int s = 0; s < 27; ++s) w2d[s] = a.pack[RF_OFF_W2D + s * 64 + lane];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.pack + RF_OFF_W1Q);
        f32x4* dst = reinterpret_cast<f32x4*>(w1q);
        for (int i = tid; i < 5120 / 4; i += RF_NT) dst[i] = src[i];
        for (int i = tid; i < 144; i += RF_NT) lds[RB_OFF_ONES + i] = i < 72 ? 1.f : 0.f;
    }
    int z_ch[RF_ZL], z_j[RF_ZL];
#pragma unroll
    for (int i = 0; i < RF_ZL; ++i) {
        int e = tid + RF_NT * i;
        if (e >= RF_NZ) e = RF_NZ - 1;
        z_ch[i] = e / 66;
        z_j[i] = e - z_ch[i] * 66;
    }
    auto z_load = [&](int rr, float* v) {
        const bool rok = rr >= 0 && rr < g.H;
        const int rc = rok ? rr : 0;
#pragma unroll
        for (int i = 0; i < RF_ZL; ++i) {
            const int col = z_j[i] - 1;
            const bool ok = rok && col >= 0 && col < g.W;
            const float x = zb[(size_t)z_ch[i] * plane + (size_t)rc * g.W + (ok ? col : 0)];
            v[i] = ok ? x : 0.f;
        }
    };
    auto z_store = [&](int rr, const float* v) {
        const int s = rf_slot4(rr);
#pragma unroll
        for (int i = 0; i < RF_ZL; ++i) zrow[(z_ch[i] * 4 + s) * RF_ZS + z_j[i]] = v[i];
    };
    // dl'(rr, col) = do[q]((rr-2)/2, (col-2)/2) at even rr, col (zero-inserted gradient of o), staged per row as
    // [q][slot][j = col + 1]; element e = tid + 128 i -> (q, j)
    int d_q[RF_DLL], d_j[RF_DLL], d_oj[RF_DLL];
    bool d_ok[RF_DLL];
#pragma unroll
    for (int i = 0; i < RF_DLL; ++i) {
        int e = tid + RF_NT * i;
        const bool live = e < RF_DL_N;
        if (!live) e = RF_DL_N - 1;
        d_q[i] = e / 66;
        d_j[i] = e - d_q[i] * 66;
        const int col = d_j[i] - 1;
        d_oj[i] = (col - 2) >> 1;
        d_ok[i] = live && d_q[i] < g.Q && col >= 2 && (col & 1) == 0 && d_oj[i] < g.OW;
    }
    float db2[RF_DLL];
#pragma unroll
    for (int i = 0; i < RF_DLL; ++i) db2[i] = 0.f;
    auto dl_load = [&](int rr, float* v) {
        const int oi = (rr - 2) >> 1;
        const bool rok = rr >= 2 && (rr & 1) == 0 && oi < g.OH;
#pragma unroll
        for (int i = 0; i < RF_DLL; ++i) {
            const bool ok = rok && d_ok[i];
            const float x = a.d_o[(((size_t)b * g.Q + (ok ? d_q[i] : 0)) * g.OH + (ok ? oi : 0)) * g.OW + (ok ? d_oj[i] : 0)];
            v[i] = ok ? x : 0.f;
        }
    };
    auto dl_store = [&](int rr, const float* v) {
        const bool own = rr >= y0 && rr < y1;
#pragma unroll
        for (int i = 0; i < RF_DLL; ++i) {
            if (tid + RF_NT * i < RF_DL_N) {
                const int o = (d_q[i] * 4 + rf_slot4(rr)) * RF_ZS + d_j[i];
                dlt[o] = v[i];
                dlo[o] = own ? v[i] : 0.f;
                db2[i] += own ? v[i] : 0.f;
            }
        }
    };
    {
        float v0[RF_ZL], v1[RF_ZL], v2[RF_ZL], e0[RF_DLL], e1[RF_DLL], e2[RF_DLL];
        z_load(y0 - 2, v0); z_load(y0 - 1, v1); z_load(y0, v2);
        dl_load(y0 - 2, e0); dl_load(y0 - 1, e1); dl_load(y0, e2);
        z_store(y0 - 2, v0); z_store(y0 - 1, v1); z_store(y0, v2);
        dl_store(y0 - 2, e0); dl_store(y0 - 1, e1); dl_store(y0, e2);
    }
    __syncthreads();

    int w1b_off[5], w1b_ky[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int q = nt * 32 + n;
        if (q < 144) {
            const int ci = q / 9, tap = q - ci * 9;
            w1b_off[nt] = RB_OFF_Z + ci * 4 * RF_ZS + tap % 3;
            w1b_ky[nt] = tap / 3;
        } else {
            w1b_off[nt] = q == 144 ? RB_OFF_ONES : RB_OFF_ONES + 72;
            w1b_ky[nt] = 3;
        }
    }
    int w2b_off[2], w2b_ky[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ct = t * 32 + n;
        if (ct < C9) {
            const int c = ct / 9, tap = ct - c * 9;
            w2b_off[t] = RB_OFF_DLO + c * 4 * RF_ZS + 2 - tap % 3;
            w2b_ky[t] = tap / 3;
        } else {
            w2b_off[t] = RB_OFF_ONES + 72;
            w2b_ky[t] = 3;
        }
    }
    const int kh = gp, px0 = kh * 32 + hh;
    f32x16 dW1acc[5], dW2acc[2];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) dW1acc[nt] = (f32x16){0};
    dW2acc[0] = dW2acc[1] = (f32x16){0};
    float V[3][8][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) V[i][c][k] = 0.f;
    const int ox = tid & 63, ocq = tid >> 6;

    // HS: this lane's 16 channels of hidden row rr at its pixel, requested one row ahead (zero outside the frame, as the
    // forward pass masked them)
    float hcur[HS ? 16 : 1], hnext[HS ? 16 : 1];
    auto h_load = [&](int rr, float* v) {
        if constexpr (HS) {
            const int hcol = gp * 32 + n;
            const bool ok = rr >= 0 && rr < g.H && hcol < g.W;
            const float* hp = a.hsaved + ((size_t)b * RF_HID + 4 * hh) * plane + (size_t)(ok ? rr : 0) * g.W + (ok ? hcol : 0);
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
        float zv[RF_ZL], dv4[RF_DLL];
        z_load(r + 2, zv);
        dl_load(r + 2, dv4);
        h_load(r + 1, hnext);
        const int ro0 = rf_slot4(r - 1) * RF_ZS, ro1 = rf_slot4(r) * RF_ZS, ro2 = rf_slot4(r + 1) * RF_ZS;
        f32x16 hacc = {0}, dh = {0};
        if (row_in) {
            if constexpr (!HS) {
            const float* zb0 = zrow + hh * 4 * RF_ZS + gp * 32 + n;
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
                    const float* zc = zb0 + (cp + 1) * 2 * 4 * RF_ZS;
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
            float bd[27];
#pragma unroll
            for (int s = 0; s < 27; ++s) {
                const int ct = 2 * s + hh;
                const int c = ct / 9, tap = ct - c * 9, ky = tap / 3, kx = tap - ky * 3;
                const int ro = ky == 0 ? ro2 : (ky == 1 ? ro1 : ro0);
                const int off = ct < C9 ? RB_OFF_DLT + c * 4 * RF_ZS + ro + 2 - kx + gp * 32 + n : RB_OFF_ONES + 72;
                bd[s] = lds[off];
            }
#pragma unroll
            for (int s = 0; s < 27; ++s) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(w2d[s], bd[s], dh, 0, 0, 0);
            const int hcol = gp * 32 + n;
            const bool colok = hcol < g.W;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float hv, dv;
                if constexpr (HS) {
                    hv = hcur[j];           // (leaky_relu keeps the sign: the saved activation is its own gate)
                    dv = colok ? dh[j] * (hv > 0.f ? 1.f : 0.01f) : 0.f;
                } else {
                    const float pre = hacc[j] + b1r[j];
                    hv = colok ? (pre > 0.f ? pre : 0.01f * pre) : 0.f;
                    dv = colok ? dh[j] * (pre > 0.f ? 1.f : 0.01f) : 0.f;
                }
                hacc[j] = hv;
                dh[j] = dv;
                const int o = rf_acc_row(j, hh) * 65 + gp * 32 + n;
                hT[o] = hv;
                dhT[o] = row_own ? dv : 0.f;
            }
        }
        if constexpr (HS) {
#pragma unroll
            for (int j = 0; j < 16; ++j) hcur[j] = hnext[j];
        }
        z_store(r + 2, zv);
        dl_store(r + 2, dv4);
        __syncthreads();

        if (row_in) {
            {
                const float* ap = dhT + n * 65 + px0;
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
                const float* ap = hT + n * 65 + px0;
                int bo[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    bo[t] = w2b_off[t] + (w2b_ky[t] == 0 ? ro2 : (w2b_ky[t] == 1 ? ro1 : (w2b_ky[t] == 2 ? ro0 : 0))) + px0;
                float av[16], b0[16], b1v[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    av[s] = ap[2 * s];
                    b0[s] = lds[bo[0] + 2 * s];
                    b1v[s] = lds[bo[1] + 2 * s];
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    dW2acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b0[s], dW2acc[0], 0, 0, 0);
                    dW2acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b1v[s], dW2acc[1], 0, 0, 0);
                }
            }
            {
                f32x16 Qt[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) Qt[t] = (f32x16){0};
                const float* qa = w1q + lane;
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
                    for (int t = 0; t < 5; ++t) Qt[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], dh[j], Qt[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 5; ++t) av[t] = an[t];
                }
#pragma unroll
                for (int jj = 0; jj < 72; ++jj) {
                    const int cl = jj / 9, tap = jj % 9, ky = tap / 3, kx = tap % 3;
                    V[ky][cl][kx] += Qt[jj / 16][jj % 16];
                }
            }
        }
        const int yo = r - 1;
        const bool fin = yo >= y0 && yo < y1;
        __syncthreads();                    // the [channel][pixel] images are consumed: vz reuses their space
        if (fin) {
#pragma unroll
            for (int cl = 0; cl < 8; ++cl)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) vz[((8 * hh + cl) * 3 + kx) * 64 + gp * 32 + n] = V[0][cl][kx];
        }
#pragma unroll
        for (int cl = 0; cl < 8; ++cl)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                V[0][cl][kx] = V[1][cl][kx];
                V[1][cl][kx] = V[2][cl][kx];
                V[2][cl][kx] = 0.f;
            }
        __syncthreads();
        if (fin) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = ocq * 8 + i;
                float s = 0.f;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xs = ox + 1 - kx;
                    if (xs >= 0 && xs < 64) s += vz[(ci * 3 + kx) * 64 + xs];
                }
                if (ox < g.W) a.dz[((size_t)b * RF_LAT + ci) * plane + (size_t)yo * g.W + ox] = s;
            }
        }
        __syncthreads();                    // vz read: the next row may overwrite the images
    }

    float* part = a.part + (size_t)wi * RF_BWD_PART;
    float* xb = hT;
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        if (kh == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) xb[rf_acc_row(j, hh) * 65 + n] = dW1acc[nt][j];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = rf_acc_row(j, hh);
                part[co * 160 + nt * 32 + n] = dW1acc[nt][j] + xb[co * 65 + n];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (kh == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) xb[rf_acc_row(j, hh) * 65 + n] = dW2acc[t][j];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ch = rf_acc_row(j, hh);
                part[32 * 160 + ch * 64 + t * 32 + n] = dW2acc[t][j] + xb[ch * 65 + n];
            }
        }
        __syncthreads();
    }
    float* cs = dhT;                        // [6][66] column sums of the owned gradient
#pragma unroll
    for (int i = 0; i < RF_DLL; ++i)
        if (tid + RF_NT * i < RF_DL_N) cs[d_q[i] * 66 + d_j[i]] = db2[i];
    __syncthreads();
    if (tid < 8) {
        float s = 0.f;
        if (tid < g.Q) for (int j = 0; j < 66; ++j) s += cs[tid * 66 + j];
        part[32 * 160 + 32 * 64 + tid] = s;
    }
}

// fixed-order sum of the workgroup slabs into the nn.Conv2d gradient layouts
__global__ void __launch_bounds__(256) reward_fused_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int Q,
                                                                        int accumulate, float* __restrict__ dw1,
                                                                        float* __restrict__ db1, float* __restrict__ dw2,
                                                                        float* __restrict__ db2) {
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, q8 = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int n1 = 32 * 144, nb1 = 32, n2 = 32 * 9 * Q, nb2 = Q;
    const bool live = i < n1 + nb1 + n2 + nb2;
    int src = 0;
    if (i < n1) { const int co = i / 144, k = i - co * 144; src = co * 160 + k; }
    else if (i < n1 + nb1) src = (i - n1) * 160 + 144;
    else if (i < n1 + nb1 + n2) { const int k = i - n1 - nb1, ch = k / (9 * Q), ct = k - ch * 9 * Q; src = 32 * 160 + ch * 64 + ct; }
    else if (live) src = 32 * 160 + 32 * 64 + (i - n1 - nb1 - n2);
    const int per = (nparts + 7) / 8, p0 = q8 * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    if (live) for (int p = p0; p < p1; ++p) s += part[(size_t)p * RF_BWD_PART + src];
    red[q8][e] = s;
    __syncthreads();
    if (q8 != 0 || !live) return;
    s = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
    float* d;
    if (i < n1) {
        const int co = i / 144, k = i - co * 144;
        d = dw1 + co * 144 + k;                                    // (32, 16, 3, 3): co, ci*9 + tap
    } else if (i < n1 + nb1) {
        d = db1 + (i - n1);
    } else if (i < n1 + nb1 + n2) {
        const int k = i - n1 - nb1, ch = k / (9 * Q), ct = k - ch * 9 * Q, q = ct / 9, tap = ct - q * 9;
        d = dw2 + (q * RF_HID + ch) * 9 + tap;                     // (3R, 32, 3, 3)
    } else {
        d = db2 + (i - n1 - nb1 - n2);
    }
    *d = accumulate ? *d + s : s;
}

extern "C" int scm_reward_fused_workspace(int B, int H, int W, int Q, size_t* bwd_floats) {
    RewGeom g;
    SCM_REQUIRE(B > 0 && Q >= 1 && Q <= 6 && rf_geom(B, H, W, Q, 512, &g), "scm_reward_fused_workspace: unsupported shape");
    if (bwd_floats) *bwd_floats = (size_t)B * g.nbands * RF_BWD_PART;
    return SCM_OK;
}

extern "C" int scm_reward_fused_fwd(const float* z, const float* pack, float* o, float* h_save, int B, int H, int W, int Q,
                                    void* stream) {
    SCM_REQUIRE(z && pack && o && B > 0 && Q >= 1 && Q <= 6, "scm_reward_fused_fwd: bad arguments");
    RewFwdArgs a;
    a.z = z; a.pack = pack; a.o = o; a.hsave = h_save;
    if (!rf_geom(B, H, W, Q, 1024, &a.g)) {
        scm_set_error("scm_reward_fused_fwd: frames must be 5..64 wide and at least 5 high (got %dx%d)", H, W);
        return SCM_EUNSUPPORTED;
    }
    const size_t lds = RF_FWD_LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(reward_fused_fwd_kernel), attr_done, (int)lds)) return rc;
    hipLaunchKernelGGL(reward_fused_fwd_kernel, dim3(B * a.g.nbands), dim3(RF_NT), lds, (hipStream_t)stream, a);
    SCM_LAUNCH_CHECK("reward_fused_fwd_kernel");
    return SCM_OK;
}

extern "C" int scm_reward_fused_bwd(const float* z, const float* pack, const float* d_o, const float* h_saved, float* dz,
                                    float* d_conv1_weight, float* d_conv1_bias, float* d_conv2_weight, float* d_conv2_bias,
                                    float* workspace, int B, int H, int W, int Q, int accumulate, void* stream) {
    SCM_REQUIRE(z && pack && d_o && dz && d_conv1_weight && d_conv1_bias && d_conv2_weight && d_conv2_bias && workspace &&
                    B > 0 && Q >= 1 && Q <= 6, "scm_reward_fused_bwd: bad arguments");
    RewBwdArgs a;
    a.z = z; a.pack = pack; a.d_o = d_o; a.dz = dz; a.part = workspace; a.hsaved = h_saved;
    if (!rf_geom(B, H, W, Q, 512, &a.g)) {
        scm_set_error("scm_reward_fused_bwd: frames must be 5..64 wide and at least 5 high (got %dx%d)", H, W);
        return SCM_EUNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = RB_LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    static std::atomic<uint64_t> attr_done_hs{0};
    const int wgs = B * a.g.nbands;
    if (h_saved) {
        if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(reward_fused_bwd_kernel<1>), attr_done_hs, (int)lds)) return rc;
        hipLaunchKernelGGL(reward_fused_bwd_kernel<1>, dim3(wgs), dim3(RF_NT), lds, s, a);
    } else {
        if (int rc = scm_ensure_lds_attr(reinterpret_cast<const void*>(reward_fused_bwd_kernel<0>), attr_done, (int)lds)) return rc;
        hipLaunchKernelGGL(reward_fused_bwd_kernel<0>, dim3(wgs), dim3(RF_NT), lds, s, a);
    }
    SCM_LAUNCH_CHECK("reward_fused_bwd_kernel");
    const int total = 32 * 144 + 32 + 32 * 9 * Q + Q;
    hipLaunchKernelGGL(reward_fused_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, s, workspace, wgs, Q, accumulate,
                       d_conv1_weight, d_conv1_bias, d_conv2_weight, d_conv2_bias);
    SCM_LAUNCH_CHECK("reward_fused_wgrad_reduce_kernel");
    return SCM_OK;
}
