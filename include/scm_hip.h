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
 *   out  = act( acc * (*scale) + bias[co] + sbias[b][co] ) (+ addend)
 *          scale (device scalar, e.g. 1/sigma of SpectralNorm), bias, sbias
 *          and addend are optional (NULL).
 *   The first co1 output channels go to y1[B,co1,H,W], the remaining
 *   cout-co1 to y2[B,cout-co1,H,W] (y2 may be NULL when co1 == cout); co1
 *   must be a multiple of 32 when y2 is used.
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
 * Pack layout: [group g][ci][tap][32 lanes][CT] with cout = g*32*CT + t*32 + lane
 * at element t, CT = min(4, ceil(cout/32)); missing couts are zero.
 */
int scm_pack_conv3x3_weights(const float* w, int cin, int cout, int transposed_src,
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
 * nn.ConvTranspose2d layout dw[ci][co][2-ky][2-kx].  accumulate = 1 adds into
 * dw / db instead of overwriting.  part_ws / csum_ws are caller-provided
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
} scm_conv3x3_wgrad_desc;

int scm_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout, size_t* part_floats,
                                size_t* csum_floats);
int scm_conv3x3_wgrad(const scm_conv3x3_wgrad_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCM_HIP_H */
