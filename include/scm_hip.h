/*
 * scm_hip.h -- C ABI of libscmgan_hip.so, the MI355X (gfx950 / CDNA4) native
 * implementation of scm-gan's action-conditional frame-prediction hot path.
 *
 * The reference (LilJing/scm-gan) has no FFI: its boundary for this path is the
 * set of PyTorch ops its modules call.  Every entry point below names the
 * reference lines whose arithmetic it replaces.  All tensors are fp32, NCHW,
 * contiguous, resident in device (HBM) memory and owned by the caller; the
 * library allocates nothing persistent, keeps no global mutable state, and
 * enqueues its kernels on the `stream` argument (a hipStream_t passed as
 * void*; NULL = the default stream).
 *
 * Error convention: every entry point returns 0 on success, a negative
 * SCM_E* code otherwise; scm_last_error() returns a thread-local message.
 * No exception crosses this boundary and nothing aborts.
 */
#ifndef SCM_HIP_H
#define SCM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCM_OK 0
#define SCM_EINVAL (-1)     /* bad shape / null pointer / unsupported flag */
#define SCM_EUNSUPPORTED (-2)
#define SCM_EHIP (-3)       /* a HIP runtime call failed */

/* padding of the 3x3 stride-1 same-size convolution */
#define SCM_PAD_ZERO 0      /* nn.Conv2d(padding=1): models.py:129-134, 260-266 */
#define SCM_PAD_CIRCULAR 1  /* padding_mode='circular', 1 px/side: models.py:51-56 */

/* activation fused in an epilogue, or its derivative applied on load */
/* MFMA operand precision of the opt-in 16-bit conv variant (scm_conv3x3_h16): products
 * of fp16 / bf16 roundings of the fp32 tensors, accumulated in fp32 */
#define SCM_PREC_F32 0
#define SCM_PREC_F16 1
#define SCM_PREC_BF16 2
/* fp32 arithmetic on the bf16 MFMA pipe: operands split exactly into three bf16 values, six partial products per
 * product, fp32 accumulation (scm_conv3x3_x3 / scm_conv3x3_wgrad_x3): fp32-grade results, not a 16-bit mode */
#define SCM_PREC_BF16X3 3

#define SCM_ACT_NONE 0
#define SCM_ACT_LRELU 1     /* F.leaky_relu(x), slope 0.01: models.py:77,146,... */
#define SCM_ACT_SIGMOID 2   /* torch.sigmoid: models.py:103,155 */

int scm_version(void);
const char* scm_last_error(void);

/* ------------------------------------------------------------------------
 * 3x3 stride-1 same-size convolution on the MFMA pipe (fp32 in, fp32 acc).
 *
 * One descriptor serves the forward conv and, with the dgrad weight pack,
 * the input-gradient conv.  Replaces nn.Conv2d.forward / ConvTranspose2d
 * forward and their autograd backward at models.py:76-102 (Transition),
 * 145-155 (Encoder), 274-278 (Decoder).
 *
 *   in   = concat(x1[B,c1,H,W], x2[B,c2,H,W]) along channels (x2 may be NULL,
 *          c2 = 0): the torch.cat of models.py:95,101 is never materialised.
 *   gate = optional tensors of the same shapes as x1/x2: when gate_act != 0
 *          every loaded input value is multiplied by act'(gate) -- this is
 *          how a backward pass folds d(leaky_relu)/d(sigmoid) into the load
 *          of dy (gate = the saved forward output).
 *   acc  = conv3x3(in, W) with `pad_mode`; W is given as a weight pack made
 *          by scm_pack_conv3x3_weights (layout documented there).
 *   out  = act( acc * (*scale) + bias[co] + sbias[b][co] )
 *          scale (device scalar, e.g. 1/sigma of SpectralNorm), bias and sbias
 *          are optional (NULL).
 *   The first co1 output channels go to y1[B,co1,H,W], the remaining
 *   cout-co1 to y2[B,cout-co1,H,W] (y2 may be NULL when co1 == cout): this is
 *   how the input-gradient conv of a two-input layer writes dx1 and dx2.
 * ---------------------------------------------------------------------- */
typedef struct scm_conv3x3_desc {
    const float* x1; const float* x2; int32_t c1; int32_t c2;
    const float* gate1; const float* gate2; int32_t gate_act;
    const float* wpack;      /* from scm_pack_conv3x3_weights */
    int32_t cout;            /* logical output channels */
    const float* scale;      /* device float, may be NULL */
    const float* bias;       /* [cout], may be NULL */
    const float* sbias;      /* [B][cout], may be NULL */
    float* y1; float* y2; int32_t co1;
    int32_t B; int32_t H; int32_t W;
    int32_t pad_mode; int32_t act;
    /* Optional 1-bit form of the leaky-relu gate (both may be NULL), words [B][ceil(C/32)][H*W],
     * bit j of word w = "channel 32w+j of the activation output is > 0":
     *   gate_mask_out : written by a forward call with act == SCM_ACT_LRELU and cout > 32
     *                   (C = cout), next to y;
     *   gate_mask_in  : read by a gradient call with gate_act == SCM_ACT_LRELU instead of
     *                   gate1 (C = c1, c2 must be 0) by the kernels for cout > 16; others
     *                   keep reading gate1, which must still be passed.
     * A gradient convolution then loads one word per pixel and 32 channels instead of 32 floats. */
    uint32_t* gate_mask_out; const uint32_t* gate_mask_in;
    /* != 0: y1 += out instead of y1 = out (y2 is always overwritten).  An activation with two consumers (the U-net
     * skips of models.py:78,82 feeding conv2/conv6 and conv3/conv5) gets the second consumer's input gradient added
     * in the epilogue of that consumer's gradient convolution instead of by a separate elementwise add. */
    int32_t accumulate_y1;
    /* scm_conv3x3_wino only (the other entry points refuse them): the 1-bit leaky-relu gate of the activation whose
     * gradient y1 (y2) is -- words [B][ceil(co1/32)][H*W] ([B][ceil((cout-co1)/32)][H*W]), the gate_mask_out of the layer
     * that PRODUCED that activation.  When given, every value written (or added) to y1 / y2 is multiplied by the
     * leaky-relu slope of its own element (1 or 0.01): an input-gradient convolution then hands the layer below a
     * gradient that is already gated, and that layer's own gradient kernels run with gate_act = 0 -- the gate is
     * applied once, by the producer's epilogue, instead of twice, by the staging code of both consumers. */
    const uint32_t* gate_mask_y1; const uint32_t* gate_mask_y2;
    /* scm_conv3x3_h16 only (the other entry points refuse them).  x_half != 0: x1 / x2 point to IEEE-half tensors (same
     * NCHW shapes), as written by a call with y_half != 0: in the 16-bit operand mode the activations between two
     * convolutions then cross HBM at two bytes per element -- the operands are rounded to fp16 once either way, when
     * stored instead of when staged.  Gradient tensors (a call's x1 in the input-gradient direction, its y1 / y2)
     * stay fp32: y_half requires accumulate_y1 == 0. */
    int32_t x_half; int32_t y_half;
} scm_conv3x3_desc;

int scm_conv3x3(const scm_conv3x3_desc* d, void* stream);

/* Number of floats in a weight pack for (cin, cout). */
size_t scm_conv3x3_pack_floats(int cin, int cout);

/*
 * Build weight packs from a torch-layout weight.
 *   w        : [cout][cin][3][3]  (nn.Conv2d.weight)          when transposed_src == 0
 *              [cin][cout][3][3]  (nn.ConvTranspose2d.weight, used as the
 *              equivalent conv: taps flipped)                   when transposed_src == 1
 *   fwd_pack : pack for y = conv(x, w)          (scm_conv3x3_pack_floats(cin, cout) floats) or NULL
 *   dgrad_pack : pack for dx = conv(dy, w^T flipped) (scm_conv3x3_pack_floats(cout, cin) floats) or NULL
 * Only input channels [w_ci0, w_ci0+cin) of a weight holding w_cin_total input
 * channels are packed (w_cin_total = 0 means cin): Transition.conv1 convolves the
 * 16 latent channels and folds its action channels into scm_action_bias.
 * Pack layout: [group g][ci][tap][32 lanes][CT] with cout = g*32*CT + t*32 + lane
 * at element t, CT = min(4, ceil(cout/32)); missing couts are zero.
 */
int scm_pack_conv3x3_weights(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0,
                             float* fwd_pack, float* dgrad_pack, void* stream);

/* ------------------------------------------------------------------------
 * Weight / bias gradient of the same convolution (autograd backward of
 * nn.Conv2d / nn.ConvTranspose2d triggered by main.py:285).
 *
 *   dyp = dy * act'(gate)            (gate = saved forward output, optional)
 *   dw[co][ci][ky][kx] (=|+=) sum_{b,y,x} dyp[b,co,y,x] * inpad[b,ci,y+ky-1,x+kx-1]
 *   csum[b][co]        =  sum_{y,x} dyp[b,co,y,x]          (optional)
 *   db[co]             (=|+=) sum_b csum[b][co]            (optional)
 * `in` is concat(x1, x2) as in scm_conv3x3.  transposed_dst = 1 writes the
 * nn.ConvTranspose2d layout dw[ci][co][2-ky][2-kx].  accumulate: bit 0 adds into
 * dw, bit 1 adds into db, instead of overwriting.  part_ws / csum_ws are caller-provided
 * workspaces sized by scm_conv3x3_wgrad_workspace (split-K partial slabs,
 * reduced in a fixed order: results are bitwise reproducible).
 * ---------------------------------------------------------------------- */
typedef struct scm_conv3x3_wgrad_desc {
    const float* x1; const float* x2; int32_t c1; int32_t c2;
    const float* dy; const float* gate; int32_t gate_act; int32_t cout;
    float* dw; float* db; float* csum;
    float* part_ws; float* csum_ws;
    int32_t B; int32_t H; int32_t W; int32_t pad_mode;
    int32_t transposed_dst; int32_t accumulate;
    int32_t dw_cin_total; int32_t dw_ci0;   /* dw holds dw_cin_total input channels (0 = c1+c2); write slice at dw_ci0 */
    const float* dot_w; float* dot_part;    /* optional: dot_part[i] = partial <dw, dot_w> of reduce block i
                                               (ceil(9*cout*cin/256) floats) for scm_sn_fold_grad */
    const uint32_t* gate_mask;              /* optional 1-bit leaky-relu gate (scm_conv3x3_desc.gate_mask_out of the
                                               forward call, C = cout); `gate` must still be passed */
    int32_t x_half;                         /* scm_conv3x3_wgrad / scm_conv3x3_wgrad_h16 on the kernels of the 16-bit
                                               operand mode: x1 / x2 are IEEE-half tensors (scm_conv3x3_desc.y_half);
                                               refused by the Winograd and split-precision kernels */
} scm_conv3x3_wgrad_desc;

int scm_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout, size_t* part_floats,
                                size_t* csum_floats);
int scm_conv3x3_wgrad(const scm_conv3x3_wgrad_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * Opt-in 16-bit-input / fp32-accumulate variant of scm_conv3x3 (no reference
 * counterpart: the reference is fp32 only; BASELINE config 5 / SURVEY.md 8(d) name
 * it as "fp16 MFMA").  Same descriptor, same fp32 NCHW tensors, same epilogue;
 * `wpack` must come from scm_pack_conv3x3_weights_h16 in the same precision.
 * The operands of every product are rounded (nearest-even) to `precision`
 * while staged; sums are fp32.  Forward uses SCM_PREC_F16, the gradient
 * convolution SCM_PREC_BF16 (fp32 exponent range: no loss scaling).
 * Pack layout: [group][chunk of 16 ci][tap][h][m][32 lanes][8 halves], element j of
 * lane l = W[co = g*32*CT + m*32 + l][ci = 16*chunk + 8*h + j][tap], zero padded.
 * ---------------------------------------------------------------------- */
size_t scm_conv3x3_pack_h16_bytes(int cin, int cout);
int scm_pack_conv3x3_weights_h16(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0,
                                 void* fwd_pack, int fwd_precision, void* dgrad_pack, int dgrad_precision,
                                 void* stream);
int scm_conv3x3_h16(const scm_conv3x3_desc* d, int precision, void* stream);

/* scm_conv3x3_wgrad with the 128 co x 64 (32) ci workgroup tiles (cout > 64, cin > 16) on 16-bit
 * operands (dyp and x rounded to `precision` while staged; fp32 accumulation, fp32 partial slabs,
 * the same fixed-order reduction, workspaces as scm_conv3x3_wgrad_workspace).  Other shapes
 * run the fp32 kernels. */
int scm_conv3x3_wgrad_h16(const scm_conv3x3_wgrad_desc* d, int precision, void* stream);

/* ------------------------------------------------------------------------
 * Spectral normalisation (spectral_normalization.py:14-68).
 *
 * scm_sn_power_iter runs `steps` consecutive SpectralNorm._update_u_v calls
 * (power_iterations = 1 each, :23-35) for up to 8 layers in one launch:
 *   v <- W^T u / (|W^T u| + 1e-12);  u <- W v / (|W v| + 1e-12);  sigma = u.(W v)
 * with W = w_bar viewed as [rows][cols]. u and v are updated IN PLACE (the
 * reference rebinds u.data / v.data); sigma[k] / inv_sigma[k] (either may be
 * NULL) receive the value after call k.
 *
 * scm_sn_fold_grad maps g = dL/d(W_bar/sigma) to the gradient of W_bar:
 *   dw_bar (=|+=) g * inv_sigma - (<g, W_bar> * inv_sigma^2) * u v^T
 * u and v are read at call time: the reference's autograd graph holds the live
 * u/v parameters, so its backward sees the latest power-iteration state for
 * every earlier forward call (oracle/scm_oracle.py sn_weight).
 * ---------------------------------------------------------------------- */
typedef struct scm_sn_layer {
    const float* w_bar; float* u; float* v; float* sigma; float* inv_sigma;
    int32_t rows; int32_t cols;
} scm_sn_layer;

int scm_sn_power_iter(int nlayers, const scm_sn_layer* layers, int steps, void* stream);
int scm_sn_fold_grad(const float* g, const float* w_bar, const float* u, const float* v, const float* inv_sigma,
                     const float* dot_part, int ndot, int rows, int cols, int accumulate, float* dw_bar,
                     void* stream);
/* dot_part/ndot: optional partial sums of <g, W_bar> (from scm_conv3x3_wgrad); when NULL the
 * dot product is computed here by one workgroup. */

/* ------------------------------------------------------------------------
 * Generic small Conv2d (square kernel K, stride, zero padding `pad`; a negative pad
 * crops: RewardPredictor.conv2 reads the interior of conv1's same-size map with pad = -1), direct
 * form on the vector ALU: RewardPredictor (models.py:230-239), Discriminator
 * (models.py:199-219), CoordConv2d's inner conv (coordconv.py:8,15).
 *   fwd:  y = act( conv(x, w) * (*scale) + bias )
 *   bwd:  dyp = dy * act'(gate);  dx = scale * conv^T(dyp, w);
 *         dw (=|+=) corr(x, dyp)  [w.r.t. the effective weight w*scale -> caller
 *         folds scale];  db (=|+=) sum dyp
 * ---------------------------------------------------------------------- */
typedef struct scm_conv2d_desc {
    const float* x; const float* w; const float* bias; const float* scale; float* y;
    const float* gate; int32_t gate_act;
    int32_t B; int32_t cin; int32_t H; int32_t W; int32_t cout;
    int32_t K; int32_t stride; int32_t pad; int32_t act;
} scm_conv2d_desc;

int scm_conv2d_fwd(const scm_conv2d_desc* d, void* stream);
/* ws: caller workspace of scm_conv2d_bwd_workspace(B, cin, cout, K) floats (needed when dw != NULL);
 * the batch is split over workgroups and reduced in a fixed order (reproducible). */
size_t scm_conv2d_bwd_workspace(int B, int cin, int cout, int K);
int scm_conv2d_bwd(const scm_conv2d_desc* d, const float* dy, float* dx, float* dw, float* db, int accumulate,
                   float* ws, void* stream);

/* ------------------------------------------------------------------------
 * Bandwidth-bound pieces.
 * ---------------------------------------------------------------------- */
/* DifferentiableBernoulliSampler forward / eval threshold (models.py:30-40,
 * 107-113). mode 0: out = (p > 0.5); mode 1: out = (noise < p); mode 2:
 * out = (philox4x32(seed, offset + *offset_dev + i/4) < p). Backward is the identity.
 * offset_dev (may be NULL): a device-resident part of the stream position -- what lets a captured iteration
 * (HIP graph, whose kernel arguments are frozen) draw fresh noise on every replay. */
int scm_bernoulli(const float* p, const float* noise, float* out, size_t n, int mode, uint64_t seed,
                  uint64_t offset, const uint64_t* offset_dev, void* stream);

/* Decoder factor sum x.view(B,L,C,H,W).sum(1) (models.py:280-287) and its
 * backward (broadcast over L). */
int scm_factor_sum(const float* x, float* y, int B, int L, int C, int HW, void* stream);
int scm_factor_bcast(const float* dy, float* dx, int B, int L, int C, int HW, void* stream);

/* per_sample[b] = mean over n = C*H*W of loss(sigmoid(logit), target).
 * kind 0: F.binary_cross_entropy with the log clamp at -100 (main.py:189-190,
 * 310-312); kind 1: squared error (main.py:814-816). prob (optional) receives
 * sigmoid(logit).  _bwd: dlogit = g[b] * d per_sample[b] / d logit. */
int scm_pixel_loss(const float* logit, const float* target, int B, int n, int kind, float* per_sample, float* prob,
                   void* stream);
int scm_pixel_loss_bwd(const float* logit, const float* target, const float* g, int B, int n, int kind,
                       float* dlogit, void* stream);

/* RewardPredictor tail (models.py:241-250): x[B][3][R][hw] -> softmax over the 3
 * classes -> p0 - p2 -> map[B][R][hw] (optional) and r[B][R] = sum over hw. */
int scm_reward_tail(const float* x, int B, int R, int hw, float* r_out, float* map_out, void* stream);
int scm_reward_tail_bwd(const float* x, const float* g_r, const float* g_map, int B, int R, int hw, float* dx,
                        void* stream);

/* Action planes of Transition.conv1 (models.py:68-76) folded into a per-sample
 * bias: sbias[b][co] = scale * sum_a act[b][a] * sum_tap w[co][c0+a][tap], w being
 * the full [cout][cin][3][3] weight.  _bwd takes csum[b][co] = d sbias and emits
 * d_act[B][A] and the action-channel slice of dw_eff (same layout as w). */
int scm_action_bias(const float* act, const float* w, const float* scale, int B, int A, int cout, int cin, int c0,
                    float* sbias, void* stream);
int scm_action_bias_bwd(const float* act, const float* w, const float* scale, const float* csum, int B, int A,
                        int cout, int cin, int c0, float* d_act, float* dw_eff, void* stream);

/* y = act(x w^T + bias): Discriminator.fc1 + leaky_relu (models.py:205,221-222). */
int scm_linear(const float* x, const float* w, const float* bias, int B, int nin, int nout, int act, float* y,
               void* stream);
int scm_linear_bwd(const float* x, const float* w, const float* y, const float* gy, int B, int nin, int nout, int act,
                   float* dx, float* dw, float* db, void* stream);

/* Loss bookkeeping of one unrolled step (main.py:178-197): mask_out = mask_in * (1 - done);
 * loss_out = loss_in + coef * rd + rec, rd_rec = { rd, rec } with rd = mean_b(mean_r((rhat-r)^2) * mask_out),
 * rec = mean_b(bce * mask_out); r / done are columns of (B,T,R) / (B,T) tensors (element strides
 * given; loss_in may be NULL).  _bwd: gradients w.r.t. rhat and bce for upstream scalar g.
 * coef_dev (may be NULL) overrides coef with a device-resident value (theta changes per iteration, a captured
 * iteration reads it from memory). */
int scm_step_loss(const float* rhat, const float* r, long r_stride, const float* done, long done_stride,
                  const float* bce, const float* mask_in, const float* loss_in, int B, int R, float coef,
                  const float* coef_dev, float* mask_out, float* loss_out, float* rd_rec, void* stream);
int scm_step_loss_bwd(const float* rhat, const float* r, long r_stride, const float* mask, const float* g, int B,
                      int R, float coef, const float* coef_dev, float* d_rhat, float* d_bce, void* stream);

/* onehot(a_idx, num_actions) (main.py:447-452) on the device: out[n][num_actions], out[i][a] = (idx[i] == a);
 * idx are int64 action indices already resident on the device (device batch interface). */
int scm_onehot(const long long* idx, long n, int num_actions, float* out, void* stream);

/* CoordConv2d input: out[B][C+2][H][W] = cat(x, coord_x, coord_y) (coordconv.py:12-14). */
int scm_coord_cat(const float* x, float* out, int B, int C, int H, int W, void* stream);

/* clip_grad_value_(clip_value) + Adam step for `count` tensors (main.py:125-129,
 * 287-296). grads are first multiplied by grad_scale (1/world_size after a sum
 * all-reduce), clamped to [-clip, clip] when clip_value > 0 (stored back, as
 * clip_grad_value_ does), then m, v, p are updated with bias correction for
 * `step` (1-based), or for *step_dev when that device pointer is given (captured iterations). The bias
 * corrections are computed on the device in both forms. Pointer tables are HOST arrays of device pointers. */
int scm_adam_clip_multi(int count, float* const* params, float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int* numel, float lr, float beta1, float beta2, float eps,
                        int step, const int* step_dev, float clip_value, float grad_scale, void* stream);

/* ------------------------------------------------------------------------
 * CSRN directional sweep (spatial_recurrent.py:55-114): per image row (along_rows
 * = 1; per column otherwise), in sweep order (reverse = 1: from the last), a
 * one-step bias-free GRU over the row's tokens -> ctx row; Conv1d(k=3, pad 1) +
 * tanh of that output is the hidden state of the next row.
 *   fwd: writes ctx[B][ch][H][W] and hprev[steps][B][ntok][ch] (state entering
 *        each step, kept for the backward pass).
 *   bwd: BPTT. dx is ACCUMULATED into (caller zero-fills; the sweeps of one CSRN
 *        add up); dparams[9*ch*ch + ch] = [d w_ih (3ch x ch), d w_hh (3ch x ch),
 *        d wc (ch x ch x 3), d bc (ch)]; slab = workspace of B * (9*ch*ch + ch).
 * ---------------------------------------------------------------------- */
typedef struct scm_csrn_desc {
    const float* x; float* ctx; float* hprev;
    const float* w_ih; const float* w_hh; const float* wc; const float* bc;
    int32_t B; int32_t ch; int32_t H; int32_t W; int32_t along_rows; int32_t reverse;
} scm_csrn_desc;

int scm_csrn_sweep_fwd(const scm_csrn_desc* d, void* stream);
int scm_csrn_sweep_bwd(const scm_csrn_desc* d, const float* dctx, float* dx, float* slab, float* dparams,
                       void* stream);

/* ------------------------------------------------------------------------
 * fp32 convolution emulated on the bf16 MFMA pipe (opt-in, csrc/conv3x3_x3.hip).  Same descriptor, tensors (fp32
 * NCHW), epilogue and gate handling as scm_conv3x3; every fp32 operand is split EXACTLY into hi + mid + lo bf16 values
 * and each product is the sum of the six partial products of order <= 2 (error <= 2^-23 of the product, i.e. one fp32
 * rounding), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Frames whose 512-pixel halo tile exceeds 768 elements
 * (wider than ~64) return SCM_EUNSUPPORTED: callers keep scm_conv3x3 there.
 *   pack: scm_conv3x3_pack_x3_bytes(cin, cout) bytes, built by scm_pack_conv3x3_weights_x3 (arguments as
 *         scm_pack_conv3x3_weights).
 * ---------------------------------------------------------------------- */
size_t scm_conv3x3_pack_x3_bytes(int cin, int cout);
int scm_pack_conv3x3_weights_x3(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0,
                                void* fwd_pack, void* dgrad_pack, void* stream);
int scm_conv3x3_x3(const scm_conv3x3_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * The same convolution (same descriptor, same epilogue contract as scm_conv3x3) evaluated by Winograd
 * F(2x2, 3x3) on the f32 MFMA pipe: Y = A^T [ (G g G^T) .* (B^T d B) ] A, 16 products per 2x2 output patch
 * and (ci, co) pair instead of 36, fp32 operands and fp32 accumulation throughout (transform entries 0, +-1,
 * +-1/2).  Replaces the same reference lines as scm_conv3x3 for the wide layers: models.py:78-95 (Transition
 * conv2..conv5), :147-152 (Encoder conv2 / conv3) and their autograd input gradients.
 *   wpack : from scm_pack_conv3x3_weights_wino -- the filters already transformed (U = G g G^T), laid out
 *           [cout group of 64][K chunk of 8 ci][16 positions][2][64][4] (scm_conv3x3_pack_wino_floats floats);
 *   scm_conv3x3_wino_serves(H, W): 1 when the kernel takes this frame size (any H, W >= 2 whose 64-tile halo
 *           rows fit its LDS staging map: up to 128 x 128), else 0 -- callers then use scm_conv3x3.
 * ---------------------------------------------------------------------- */
size_t scm_conv3x3_pack_wino_floats(int cin, int cout);
int scm_pack_conv3x3_weights_wino(const float* w, int cin, int cout, int transposed_src, int w_cin_total, int w_ci0,
                                  float* fwd_pack, float* dgrad_pack, void* stream);
int scm_conv3x3_wino_serves(int H, int W);
/* Layers with <= 16 output channels (Transition.conv6, Encoder.conv4, the input gradient of Transition.conv1) have their
 * own form on the 16-row MFMA: the pack functions choose its layout for such a direction, scm_conv3x3_wino takes it for
 * ungated inputs, one output tensor and no accumulate on frames scm_conv3x3_wino16_serves accepts. */
int scm_conv3x3_wino16_serves(int H, int W);
int scm_conv3x3_wino(const scm_conv3x3_desc* d, void* stream);
/* Weight / bias gradient (scm_conv3x3_wgrad's descriptor, workspaces and reduce) by the same algorithm:
 * dL/dg = G^T [ sum over 2x2 output patches of (A dY A^T) .* (B^T d B) ] G.  Serves layers with >= 48 channels on both
 * sides, W % 4 == 0 and 16-byte aligned tensors; any other call is forwarded to scm_conv3x3_wgrad. */
int scm_conv3x3_wgrad_wino(const scm_conv3x3_wgrad_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * Fused decoder + pixel loss: one pass over the latent (SURVEY.md 8(b) scm_decoder_fused_*).
 *
 * Replaces Decoder.forward (models.py:270-291: ConvTranspose2d 16->64 + leaky_relu, ConvTranspose2d
 * 64->16*C, view(B,16,C,H,W).sum(dim=1)), the torch.sigmoid + decoder_pixel_loss of the training loop
 * (main.py:188-190, 310-312) or the squared error of the eval rollout (main.py:814-816), and their autograd
 * backward (main.py:285).  The 64-channel hidden map and the 16*C-channel per-factor map never exist in
 * HBM: per frame the forward reads 16 latent planes and writes C logit planes.  latent_size 16, C <= 3.
 *
 *   pack   : scm_decoder_fused_pack_floats() floats, built by scm_decoder_fused_pack from the four
 *            ConvTranspose2d parameters: conv1.weight (16,64,3,3), conv1.bias (64), conv2.weight (64,16*C,3,3),
 *            conv2.bias (16*C); rebuild whenever they change.
 *   fwd    : logits[B][C][H][W] (optional, may be NULL when only the loss is wanted);
 *            loss_kind 0: loss[b] = mean_{c,h,w} BCE(sigmoid(logit), target) with torch's log clamp at -100,
 *            loss_kind 1: mean squared error of sigmoid(logit), loss_kind -1: no loss (target/loss may be NULL).
 *            workspace: scm_decoder_fused_workspace()'s fwd_floats (per-workgroup loss partials, summed in a
 *            fixed order: bitwise reproducible).
 * ---------------------------------------------------------------------- */
typedef struct scm_decoder_fused_desc {
    const float* z;          /* [B][16][H][W] */
    const float* pack;
    const float* target;     /* [B][C][H][W] or NULL */
    float* logits;           /* [B][C][H][W] or NULL */
    float* loss;             /* [B] or NULL */
    float* workspace;
    int32_t B; int32_t H; int32_t W; int32_t C;
    int32_t loss_kind;
    float* h_save;           /* optional [B][64][H][W]: the forward keeps its hidden map (leaky_relu applied, zero outside the
                              * image) for scm_decoder_fused_bwd, which then does not recompute it; NULL: not written */
} scm_decoder_fused_desc;

size_t scm_decoder_fused_pack_floats(void);
int scm_decoder_fused_pack(const float* conv1_weight, const float* conv1_bias, const float* conv2_weight,
                           const float* conv2_bias, int latent, int C, float* pack, void* stream);
int scm_decoder_fused_workspace(int B, int H, int W, int C, size_t* fwd_floats, size_t* bwd_floats);
int scm_decoder_fused_fwd(const scm_decoder_fused_desc* d, void* stream);

/* Backward of the fused decoder + loss (autograd of models.py:270-291 + main.py:188-190 under main.py:285).
 * h is recomputed from z; dl = g_loss[b]/n * dloss/dlogit comes from the saved logits and the target.
 *   dz                    [B][16][H][W], written
 *   d_conv{1,2}_{weight,bias}: gradients in the ConvTranspose2d parameter layouts (conv2's broadcast over the 16
 *                         factors); accumulate != 0 adds into them (existing .grad buffers), else overwrites
 *   workspace             scm_decoder_fused_workspace()'s bwd_floats (per-workgroup partial slabs, fixed-order sum) */
typedef struct scm_decoder_fused_bwd_desc {
    const float* z; const float* pack; const float* logits; const float* target;
    const float* g_loss;     /* [B] upstream gradient of the per-sample loss */
    float* dz;
    float* d_conv1_weight; float* d_conv1_bias; float* d_conv2_weight; float* d_conv2_bias;
    float* workspace;
    int32_t B; int32_t H; int32_t W; int32_t C;
    int32_t loss_kind;       /* 0 BCE, 1 squared error */
    int32_t accumulate;
    const float* h_saved;    /* the forward's h_save of the same z and pack, or NULL (the hidden map is recomputed from z) */
} scm_decoder_fused_bwd_desc;

int scm_decoder_fused_bwd(const scm_decoder_fused_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * Fused reward-head convolutions: RewardPredictor.conv1 (16 -> 32, 3x3 valid) + leaky_relu + conv2 (32 -> Q = 3R,
 * 3x3, stride 2, valid), models.py:235-240, in one pass over the latent; the 32-channel hidden map stays on chip.
 * The softmax tail (models.py:241-250) is scm_reward_tail on the result.  latent_dim 16, Q <= 6, frames 5..64 wide.
 *   pack : scm_reward_fused_pack_floats() floats from conv1.weight (32,16,3,3), conv1.bias, conv2.weight (Q,32,3,3),
 *          conv2.bias (nn.Conv2d layouts); rebuild when they change.
 *   fwd  : o[B][Q][OH][OW], OH = (H-5)/2 + 1, OW = (W-5)/2 + 1.
 *   bwd  : from d_o (gradient of o): dz[B][16][H][W] and the four parameter gradients (accumulate != 0 adds into
 *          them); workspace = scm_reward_fused_workspace() floats (per-workgroup slabs, fixed-order sum).
 * ---------------------------------------------------------------------- */
size_t scm_reward_fused_pack_floats(void);
int scm_reward_fused_pack(const float* conv1_weight, const float* conv1_bias, const float* conv2_weight,
                          const float* conv2_bias, int latent, int Q, float* pack, void* stream);
int scm_reward_fused_workspace(int B, int H, int W, int Q, size_t* bwd_floats);
/* h_save (fwd, optional): [B][32][H][W], the hidden map in same-size coordinates (leaky_relu applied), kept for the
 * backward pass; h_saved (bwd, optional): that buffer -- the backward then does not recompute conv1; NULL: recomputed. */
int scm_reward_fused_fwd(const float* z, const float* pack, float* o, float* h_save, int B, int H, int W, int Q,
                         void* stream);
int scm_reward_fused_bwd(const float* z, const float* pack, const float* d_o, const float* h_saved, float* dz,
                         float* d_conv1_weight, float* d_conv1_bias, float* d_conv2_weight, float* d_conv2_bias,
                         float* workspace, int B, int H, int W, int Q, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCM_HIP_H */
