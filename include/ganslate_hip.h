/*
 * ganslate_hip.h — C ABI of libganslate_hip.so, the MI355X (gfx950) drop-in for the arithmetic of
 * ganslate's GAN training step.
 *
 * The reference (ganslate-team/ganslate @ v1) has no native code: every entry point below replaces a
 * torch / cuDNN / apex call site on the hot path `BaseGAN.optimize_parameters`
 * (ganslate/nn/gans/unpaired/cyclegan.py:92-124). The reference call site each function stands in for is
 * cited next to it. Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - every call enqueues on the caller's `stream` (a hipStream_t passed as void*) and never
 *     synchronises; the caller owns all tensors, the library owns nothing but a 256-byte zero page;
 *   - return value 0 = ok, non-zero = error, message via gs_last_error();
 *   - one process per GPU, like the reference's DDP launch: the library's state (zero page, loss-reduction workspaces per
 *     launching stream, kernel-selection options) is per process. Launch calls may come from several host threads (the
 *     autograd engine runs backward on its own): workspace assignment is locked and gs_last_error() is per thread — read
 *     it on the thread whose call failed. gs_set_option is not synchronised against concurrent launches.
 *
 * Data layout: activations are NHWC bf16 with the channel count padded to a multiple
 * of 8 ("act" below); images at the network boundary are NCHW fp32 exactly as the reference's
 * `visuals` (cyclegan.py:39); master weights/grads/Adam state are fp32 in the "OTI" layout
 * [P][T][Q] (P = out channels for Conv, in channels for ConvTranspose; T = kh*kw taps, row-major;
 * Q = the other channel count, padded to a multiple of 8); bf16 weight packs are [rows][Kp] with
 * Kp = roundup(T_class*Q, 64), K contiguous.
 */
#ifndef GANSLATE_HIP_H
#define GANSLATE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_MAX_TAPS 352   /* 7x7x7 = 343 taps of the 3-D stem / output convs (resnet3d.py:25,64) */

/* border handling of the gathered operand (reference: nn.ReflectionPad2d resnet2d.py:24,
 * nn.ReplicationPad3d resnet3d.py:15, Conv2d(padding=1) zero padding resnet2d.py:35) */
enum { GS_BORDER_ZERO = 0, GS_BORDER_REFLECT = 1, GS_BORDER_REPLICATE = 2 };
/* activations (nn.ReLU resnet2d.py:27, nn.LeakyReLU(0.2) patchgan2d.py:30, nn.Tanh resnet2d.py:65) */
enum { GS_ACT_NONE = 0, GS_ACT_RELU = 1, GS_ACT_LRELU = 2, GS_ACT_TANH = 3 };

/* One "generalised convolution" class:
 *   out[n, z*so+pz, i*so+py, j*so+px, co] = bias[co] + sum_{t<T} sum_{ci<Ci}
 *        in[n, B(z*si+dd[t]), B(i*si+dh[t]), B(j*si+dw[t]), ci] * w[co][t*Ci+ci]     (z<Dc, i<Hc, j<Wc)
 * With so=1 it is nn.Conv2d/Conv3d forward (stride si) or the data-gradient of a stride-1 conv; with so=2
 * it is one output-parity class of nn.ConvTranspose2d/3d(stride=2) forward or of the data-gradient of a
 * stride-2 conv. B() applies `border`. 2-D tensors are the depth-1 case (Di=Do=Dc=1, pz=0, dd[]=0);
 * volumes are NDHWC (resnet3d.py:25-64, patchgan3d.py:28-60). Di*Hi must stay below 32768. */
typedef struct gs_gconv_desc {
  int32_t N, Hi, Wi, Ci;       /* gathered input: logical dims; Ci multiple of 8, Ci/8 a power of two */
  int32_t Di, Do, Dc, pz;      /* depth of the input / output / class and the depth phase (1,1,1,0 in 2-D) */
  int32_t in_cs, in_co;        /* input channel stride / channel offset in elements (concat views) */
  int32_t Ho, Wo, Co;          /* full output dims; Co = number of channels written, multiple of 8 */
  int32_t out_cs, out_co;      /* output channel stride / offset */
  int32_t Hc, Wc;              /* class extent */
  int32_t so, py, px, si;      /* output stride+phase, input stride */
  int32_t T;                   /* taps in this class */
  int32_t Kp;                  /* padded K of the weight pack = roundup(T*Ci, 64) */
  int32_t w_rows;              /* rows present in the weight pack (>= Co) */
  int32_t border;              /* GS_BORDER_* */
  int32_t act;                 /* GS_ACT_* applied in the epilogue (after bias) */
  float   slope;               /* LeakyReLU slope */
  int32_t stats_slots;         /* partial-stat slots per image in `stats` (0 = no stats) */
  int32_t stats_slot0;         /* first slot this class writes; it writes ceil(Dc*Hc*Wc/tile_m) slots */
  int32_t accumulate;          /* != 0: out += result (bf16 read-modify-write; the gradient joins of the additive
                                * coupling blocks, ganslate/nn/invertible.py:8-48); no bias/activation/stats then */
  int8_t  dh[GS_MAX_TAPS];
  int8_t  dw[GS_MAX_TAPS];
  int8_t  dd[GS_MAX_TAPS];
} gs_gconv_desc;

/* Weight-gradient of the same family:
 *   dw[p][t*Q+q] += sum_{n,z,i,j} a[n,z,i,j,p] * g[n, B(z*si+dd[t]), B(i*si+dh[t]), B(j*si+dw[t]), q]
 * (`a` is the dense side: dY for Conv, X for ConvTranspose; `g` the gathered side; Da=Dg=1 in 2-D). */
typedef struct gs_wgrad_desc {
  int32_t N, Ha, Wa, P, a_cs, a_co;
  int32_t Hg, Wg, Q, g_cs, g_co;    /* Q multiple of 8, Q/8 a power of two */
  int32_t Da, Dg;
  int32_t si, T, border;
  int32_t dw_ld;                    /* leading dimension of dw rows = T*Q */
  int32_t dw_fresh;                 /* hint: the caller guarantees that dw holds zeros (first weight gradient of this layer since
                                     * the optimiser cleared the buffer): a launch that is the only contributor of its elements
                                     * stores instead of read-add-store. 0 is always valid. The hint covers the FIRST operand
                                     * pair of a call only (gs_wgrad_pair / gs_wgrad_ws with a2: the second pair always adds).
                                     * One-split launches add to dw with plain loads / stores, not atomics: two weight-gradient
                                     * launches of the same layer must not overlap on different streams (order them with an
                                     * event, as NativeNet does between the backward passes of a network). */
  int8_t  dh[GS_MAX_TAPS];
  int8_t  dw_[GS_MAX_TAPS];
  int8_t  dd[GS_MAX_TAPS];
} gs_wgrad_desc;

/* Twin batch: two networks of IDENTICAL architecture over one batch — the two generators (discriminators) of a CycleGAN
 * step see independent data in every phase (cyclegan.py:126-152: G_AB(real_A) next to G_BA(real_B), G_AB(fake_A) next to
 * G_BA(fake_B); :154-189 for D_B / D_A), and with per-sample InstanceNorm nothing couples the images of a batch. Images
 * [0, n_split) use the weights / bias / gradient buffers passed to the call, images [n_split, N) the same layouts
 * `*_delta` BYTES further on (the second network's pack, master and gradient buffers have the first one's layout). */
typedef struct gs_twin {
  int32_t n_split, pad_;
  int64_t w_delta;             /* bf16 weight pack */
  int64_t bias_delta;          /* fp32 bias vector (master buffer) */
  int64_t dw_delta;            /* fp32 weight-gradient buffer (gs_wgrad_ws_twin) */
} gs_twin;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int gs_init(int device);                 /* torch.cuda.set_device + lazy cuDNN handle (base.py:84-91) */
void gs_shutdown(void);
const char* gs_last_error(void);
/* Kernel-selection switches (A/B measurements, parity tests): the library reads NO environment variable; the host
 * side maps its GS_* variables onto these (ganslate_amd/hip/ops.py). Names: splitk, splitk_max_blocks, splitk_target,
 * hconv, hconv_wide, hconvw_persist, hstrip_regs, gconv_twin, wgrad_twin, hwgrad, hwgrad_wide, hwgrad_planes, norm_bwd_ppb, norm_apply_unroll, gconv_tile288,
 * gconv_multi, hconvw_ring, hconvt (smallest grid the parity-class halo kernel takes, 0 = off), hstrip (same for the k7
 * boundary-conv kernel); round 5 / 6: gconv_persist, hconvt_persist, wgrad_rows, splitk_multi, splitk_ring, gconv_ring4,
 * hconv5 (register-resident-weights kernel for the 16 -> 16 channel k5 volume convs: smallest volume in units of 2048 voxels,
 * 0 = off), hconv5_seg (z segments per column, 0 = auto), hwgrad2 / hconv2 (double-buffered volume forms of the narrow weight
 * gradient / forward kernels), pwise (one-tap layers with <= 8 channels on one side — the V-Net's 32 -> 1 output conv — on
 * register-operand kernels: smallest volume in units of 2048 voxels, 0 = off). Every setting computes the same function (up to the fp32 summation order and, for hconvw_ring,
 * where the bf16 rounding of the folded gradient happens); none skips work — EXCEPT the timing ablations ring_apply > 1 and
 * ring_dbg != 0, which produce wrong results by design and exist for profiles/ only. Unknown name -> non-zero. */
int gs_set_option(const char* name, int value);
int gs_get_option(const char* name, int* value);
int gs_tile_m(const gs_gconv_desc* d);   /* pixel-tile height of the im2col kernel for this class */
/* number of partial-statistics slots per image gs_gconv_forward writes for this class (pixel tiles of the im2col
 * kernel, or output boxes of the halo-resident kernel narrow stride-1 layers run on): size `stats` with it */
int gs_gconv_stat_slots(const gs_gconv_desc* d);

/* ---- convolution family (torch.nn.Conv2d / ConvTranspose2d forward + autograd backward) --------- */
/* resnet2d.py:25,35,52-57,65,80-87; patchgan2d.py:29,36-62; backward via loss.backward() base.py:170 */
int gs_gconv_forward(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                     void* out, float* stats, void* stream);
/* The output-parity classes of ONE layer (stride-2 transposed conv forward, data gradient of a stride-2 conv: 4 classes in
 * 2-D, 8 in 3-D — SURVEY.md §2.3 K3/K4; resnet2d.py:52-57, patchgan2d.py:36-48 backward) in one launch. descs[c] /
 * w_packs[c] are what `count` calls of gs_gconv_forward would take; the classes must agree in everything but taps, Kp,
 * output phase (pz,py,px) and stats_slot0, with at most 8 taps each — otherwise (and for count == 1) the call runs them one
 * after the other, so it is always valid to use. On the im2col kernel the results equal the separate launches bit for bit.
 * 2-D k3 / k4 layers with channel counts that are multiples of 64, a class grid that is a multiple of 16 and a large enough
 * grid run all four classes out of one halo-resident pass instead (hconvt.hip, option `hconvt`): same function, another
 * fp32 summation order, and of the layer's `stats_slots` slots the first (class grid / 256) hold the per-box sums over all
 * classes while the others are written as zeros — consumers sum the slots, as gs_inorm_finalize does. */
int gs_gconv_forward_multi(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                           const float* bias, void* out, float* stats, void* stream);
/* The same with split-K over the MERGED grid (unet2d.py:110-157: the U-Net's inner stride-2 transposed convs and the data
 * gradients of its inner stride-2 convs — four classes of 1-512 pixels each with K = 4 x 256..2048): (classes x tiles x
 * splits) workgroups write fp32 partial sums to ws[split][output pixel][Co] (the classes partition the output pixels) and
 * one finalize pass applies bias / activation / statistics for all classes. gs_gconv_multi_splitk_ws_floats: floats of ws
 * the call wants, 0 when it would not split (classes that do not merge, a layer the halo-resident class kernel takes, a
 * merged grid that is large enough). ws == NULL or smaller than that: exactly gs_gconv_forward_multi. Deterministic (fixed
 * summation order); per-class launches through gs_gconv_forward_ws give the same sums in another split. */
int64_t gs_gconv_multi_splitk_ws_floats(const gs_gconv_desc* const* descs, int32_t count);
int gs_gconv_forward_multi_ws(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                              const float* bias, void* out, float* stats, float* ws, int64_t ws_floats, void* stream);

/* Data-gradient launch of a stride-1 conv with the first pass of the consumer's InstanceNorm backward fused into its
 * epilogue: while the tile of the (padded-domain) gradient g is stored, the per-tile sums of ghat = (fold(g) + g2) *
 * act'(yhat), ghat * yhat and yhat over the pixels of the tile are written to partial[N][slots][3][C] (slots =
 * gs_gconv_stat_slots(d)), exactly what the reduction pass of gs_inorm_act_backward would produce — pass that buffer to it
 * as `scratch` with pre_slots = slots. y / mean_rstd: raw output and statistics of the conv in front of that norm on the
 * unpadded domain Dy x Hy x Wy; the launch's output domain is that domain padded by `fold` (resnet2d.py:80-87 backward). */
typedef struct gs_gconv_fuse {
  const void* y;
  const float* mean_rstd;
  const void* g2;              /* optional residual-join gradient on the unpadded domain */
  float* partial;
  int32_t Dy, Hy, Wy;
  int32_t fold, fold_mode, act;
  float slope;
} gs_gconv_fuse;
int gs_gconv_forward_fused(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                           float* stats, const gs_gconv_fuse* fuse, void* stream);

/* The fused data gradient (contract of gs_gconv_forward_fused: sums over the consumer's InstanceNorm-backward terms in the
 * epilogue, one slot of [3][Co] floats per workgroup tile) for the output-parity classes of a stride-2 conv's data gradient
 * (resnet2d.py:35 backward) when the layer runs on the halo-resident class kernel: gs_gconv_multi_fused_slots says whether
 * (0 = no) and how many slots per image the launch writes; fuse->fold must be 0 (zero-padded layers). */
int gs_gconv_multi_fused_slots(const gs_gconv_desc* const* descs, int32_t count);
int gs_gconv_forward_multi_fused(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                                 void* out, const gs_gconv_fuse* fuse, void* stream);
/* Unpadded form of the same launch for the reflect-padded (pad 1) wide 3x3 residual convs (resnet2d.py:80-87): when
 * gs_gconv_ring_slots(d) > 0 for the ZERO-border data-gradient descriptor on the unpadded domain (Hi x Wi = Ho x Wo = the
 * conv's input extent, taps t = 3*ry + rx at (dh, dw) = (1 - ry, 1 - rx)), gs_gconv_forward_fused may be called with that
 * descriptor and fold = 1, fold_mode = reflect, Hy x Wy = Ho x Wo: the launch computes the gradient pixels one step
 * outside the image as well and adds them to the pixels the reflection folds them onto (in fp32, before rounding), so
 * `out` is the finished input gradient and the consumer runs with fold = 0. partial is [N][slots][3][C] with
 * slots = gs_gconv_ring_slots(d). Returns 0 slots when the layer / grid does not suit the halo kernel; the padded form
 * above is always available. */
int gs_gconv_ring_slots(const gs_gconv_desc* d);
/* The same launch with the consumer's WHOLE InstanceNorm backward inside it (round 5): where gs_inorm_act_backward(pre_slots)
 * would follow the ring-form launch — ganslate/nn/generators/resnet/resnet2d.py:80-93 backward: conv <- ReflectionPad <-
 * [ReLU <-] InstanceNorm2d (ganslate/nn/utils.py:53-59) — the boxes of an image meet inside the launch (their partial sums
 * written through to memory, one arrival counter per image and channel tile), add the slots up in slot order and write
 * dy = rstd * (ghat - mean ghat - yhat * mean(ghat * yhat)) instead of the input gradient; with a residual-join gradient
 * fuse->g2 and total != NULL also the total gradient gx + g2 the skip path wants. Bit for bit the results of the two launches;
 * fuse->partial keeps its layout ([N][slots][3][C] partial sums, then [N][3][C] per-image totals for gs_norm_bias_grads).
 * gs_gconv_ring_apply_words(d): int32 words of the rendezvous buffer `sync` (zero-filled once by the caller, left zero by every
 * launch; one buffer per stream), 0 when the launch cannot run in this form — every workgroup must be resident at once, so
 * only persistent grids of at most one workgroup per CU qualify, and the caller must not run two such launches concurrently
 * on different streams. */
int gs_gconv_ring_apply_words(const gs_gconv_desc* d);
int gs_gconv_ring_apply(const gs_gconv_desc* d, const void* in, const void* w_pack, const gs_gconv_fuse* fuse, void* dy,
                        void* total, int32_t* sync, const gs_twin* tw, void* stream);
/* Twin batches (gs_twin above). gs_gconv_twin_native: 1 when the kernel gs_gconv_forward (fuse == NULL) or
 * gs_gconv_forward_fused (fuse != NULL) would pick for `d` — d->N = the whole batch of both networks — selects the weight
 * set per image; 0: the caller runs the two halves as two launches (always possible: the halves are contiguous).
 * gs_gconv_forward_twin is that launch (fuse == NULL: gs_gconv_forward, else gs_gconv_forward_fused) and fails where
 * gs_gconv_twin_native says 0. The wide 3x3 residual convs (resnet2d.py:80-87) are native: a twin launch at batch 2 x 8 has
 * 512 tiles for 256 CUs and every workgroup walks two of them, the second tile's operands arriving under the first one's
 * K loop (csrc/hconvw.hip). */
int gs_gconv_twin_native(const gs_gconv_desc* d, const gs_gconv_fuse* fuse);
/* The same for the output-parity classes of one layer (gs_gconv_forward_multi / _multi_fused): native where the classes run on
 * the halo-resident class kernel (2-D k3 / k4 stride-2 layers with 64-multiple channels, csrc/hconvt.hip), which picks the
 * packs per 16 x 16 box; the descriptors carry the whole batch of both networks. fuse == NULL: bias / stats as in
 * gs_gconv_forward_multi; else the contract of gs_gconv_forward_multi_fused (bias = stats = NULL). */
int gs_gconv_multi_twin_native(const gs_gconv_desc* const* descs, int32_t count);
int gs_gconv_forward_multi_twin(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                                const float* bias, void* out, float* stats, const gs_gconv_fuse* fuse, const gs_twin* tw,
                                void* stream);
int gs_gconv_forward_twin(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                          float* stats, const gs_gconv_fuse* fuse, const gs_twin* tw, void* stream);
/* Same launch as gs_gconv_forward with a caller-owned fp32 workspace for split-K: layers with few output tiles and a
 * long K loop (U-Net bottleneck convs unet2d.py:129-136, the PatchGAN 512->1 tail patchgan2d.py:62) run their K range
 * split over the chip and a second pass sums the partial results and applies bias / statistics / activation.
 * gs_gconv_splitk_ws_floats(d) is the workspace the launch wants (0: it does not split; gs_gconv_forward_ws then
 * equals gs_gconv_forward). Results match gs_gconv_forward up to the fp32 summation order. */
int64_t gs_gconv_splitk_ws_floats(const gs_gconv_desc* d);
int gs_gconv_forward_ws(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                        void* out, float* stats, float* ws, int64_t ws_floats, void* stream);
int gs_wgrad(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, void* stream);
/* dw += wgrad(a1, g1) + wgrad(a2, g2): two operand pairs of the same layer and shapes — the two backward passes a
 * generator sees per step (cyclegan.py:139-150: G_AB(real_A) and G_AB(fake_A)) — in one launch where possible */
int gs_wgrad_pair(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2, float* dw,
                  void* stream);
/* Weight gradient + Adam in one launch, for a layer whose gradient has ONE contributor in the optimiser step (a network that
 * takes one backward pass per step — Pix2Pix's generator, pix2pix.py:84-88 — and a layer of few pixels, where the launch is one
 * workgroup per output tile): the tile's sums never go to memory; its workgroup updates the parameters, the moments and the bf16
 * pack groups right away (the arithmetic of gs_adam_step_dev_packs with grad_scale 1, element for element). The gradient buffer of
 * the layer is neither read nor written (it stays cleared). Pointers are to the LAYER's slice (element 0 = dw[0][0], a multiple of
 * 8 elements into the flat buffers, 16-byte aligned); inv_f / inv_d as in gs_adam_step_dev_packs, offset to the slice's first
 * group. gs_wgrad_adam_eligible: 1 when the layer runs as a one-split launch of the im2col kernel; else run gs_wgrad_ws and the
 * optimiser as usual. */
typedef struct gs_adam_fuse {
  float *p, *m, *v;
  const float* hyper;          /* device float[6]: lr, beta1, beta2, eps, 1 - beta1^t, sqrt(1 - beta2^t) */
  const int32_t* inv_f; void* fpack;
  const int32_t* inv_d; void* dpack;
  /* optional (tr_pack != NULL): the layer's TRANSPOSED pack (data-gradient pack of a conv, forward pack of a transposed conv:
   * 8 consecutive pack elements = 8 consecutive rows p of one (tap, q) column) written by the same launch, so that no refresh
   * from the master is needed afterwards: pack element of W[p][t][q] = tr_base[t] + q * tr_kp[t] + p (device int32[T]; tr_base[t]
   * < 0: the tap is in no class; all three multiples of 8) */
  const int32_t* tr_base; const int32_t* tr_kp; void* tr_pack;
} gs_adam_fuse;
int gs_wgrad_adam_eligible(const gs_wgrad_desc* d);
int gs_wgrad_adam(const gs_wgrad_desc* d, const void* a, const void* g, const gs_adam_fuse* adam, void* stream);
/* Deterministic form of gs_wgrad (a2 = g2 = NULL) / gs_wgrad_pair: workgroups that share output elements (split-K over
 * pixels) write partial sums to slabs of a caller-owned workspace and a second launch adds the slabs to dw in a fixed
 * order, instead of fp32 atomics on dw — two runs give bit-identical gradients (torch.use_deterministic_algorithms-like
 * behaviour of the reference's cuDNN weight gradients is NOT guaranteed either; this is what the loss-curve parity tests
 * run on). gs_wgrad_ws_floats: workspace the call wants (floats; < 0: bad descriptor; 0: every output element has a single
 * contributing workgroup, which then accumulates straight into dw — ws may be NULL). */
int64_t gs_wgrad_ws_floats(const gs_wgrad_desc* d, int32_t pair);
int gs_wgrad_ws(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2, float* dw,
                float* ws, int64_t ws_floats, void* stream);
/* Twin batches (gs_twin): d->N = all images of both networks (each operand tensor holds the first network's N / 2 images,
 * then the second's); the second network's gradient goes to dw + tw->dw_delta bytes. gs_wgrad_twin_native: 1 when the
 * layer's kernel takes both networks in one launch (the wide 3x3 residual convs, resnet2d.py:80-87 backward: half the pixel
 * splits per network, half the slab traffic); else the caller runs gs_wgrad_ws on the two halves. Workspace:
 * gs_wgrad_ws_floats_twin (both networks' slabs). Deterministic like gs_wgrad_ws. */
int gs_wgrad_twin_native(const gs_wgrad_desc* d, int32_t pair);
int64_t gs_wgrad_ws_floats_twin(const gs_wgrad_desc* d, int32_t pair);
int gs_wgrad_ws_twin(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2, float* dw,
                     float* ws, int64_t ws_floats, const gs_twin* tw, void* stream);
/* db[c] += sum over pixels of dy[pix, c]  (bias gradient of any conv) */
int gs_bias_grad(const void* dy, int64_t pixels, int32_t C, int32_t cs, int32_t co, float* db, void* stream);
/* deterministic form (per-chunk partial sums in a caller-owned workspace, added in chunk order) */
int64_t gs_bias_grad_ws_floats(int64_t pixels, int32_t C);
int gs_bias_grad_ws(const void* dy, int64_t pixels, int32_t C, int32_t cs, int32_t co, float* db, float* ws,
                    int64_t ws_floats, void* stream);
/* the same over the first 8 channels of the window, adding only db[0 .. c_valid), c_valid <= 8: a bias whose channel count is
 * not a multiple of 8 (the generators' 3-channel output conv, resnet2d.py:62-65) is followed by the next parameter in the flat
 * gradient buffer. Workspace: gs_bias_grad_ws_floats(pixels, 8) */
int gs_bias_grad_head_ws(const void* dy, int64_t pixels, int32_t cs, int32_t co, int32_t c_valid, float* db, float* ws,
                         int64_t ws_floats, void* stream);

/* ---- InstanceNorm + activation (nn.InstanceNorm2d eps=1e-5 affine=False, nn/utils.py:53-59) ------ */
/* partial [N][slots][2][C] -> mean_rstd [N][2][C] */
int gs_inorm_finalize(const float* partial, int32_t N, int32_t slots, int32_t C, int64_t hw, float eps,
                      float* mean_rstd, void* stream);
/* x = act((y-mean)*rstd) [+ res]   (resnet2d.py:26-27, 83-84, 87, 93) */
int gs_inorm_act_forward(const void* y, const float* mean_rstd, const void* res, void* x, int32_t N,
                         int64_t hw, int32_t C, int32_t act, float slope, void* stream);
/* The two calls above as ONE launch: every workgroup adds up the partial slots of its own 64 channels in its prologue,
 * the first pixel chunk's workgroups also write mean_rstd [N][2][C] (needed by the backward pass), then x is produced as
 * in gs_inorm_act_forward. (C % 64 != 0: falls back to the two launches.) */
int gs_inorm_stats_act_forward(const void* y, const float* partial, int32_t slots, float eps, float* mean_rstd,
                               const void* res, void* x, int32_t N, int64_t hw, int32_t C, int32_t act, float slope,
                               void* stream);
/* Backward of the above. g_pad is the incoming gradient on a domain padded by `fold` on each side of every
 * spatial axis with extent > 1 (the data-gradient of a reflect/replicate-padded conv; fold=0 for a plain gradient)
 * — the border is folded back here (adjoint of nn.ReflectionPad2d resnet2d.py:24 / nn.ReplicationPad3d
 * resnet3d.py:24,78). D = 1 for 2-D tensors. g2 (optional, unpadded) is added (residual join).
 * Outputs dy (gradient w.r.t. the conv output y) and optionally gsum = fold(g_pad)+g2 (skip path).
 * If mean_rstd is NULL there is no norm: `y` then holds the activation OUTPUT and dy = g*act'(y).
 * bias_grad (optional, norm only): db[c] += sum over pixels of dy — the gradient of the bias of the conv in front
 * of the norm, obtained from the reduction sums (it is identically zero up to rounding, like the reference's). */
int gs_inorm_act_backward(const void* g_pad, const void* g2, const void* y, const float* mean_rstd,
                          void* dy, void* gsum, float* scratch, float* bias_grad, int32_t N, int32_t D, int32_t H,
                          int32_t W, int32_t C, int32_t fold, int32_t fold_mode, int32_t act, float slope,
                          int32_t pre_slots, void* stream);
/* pre_slots > 0: `scratch` already holds the partial sums [N][pre_slots][3][C] (written by gs_gconv_forward_fused)
 * followed by room for the [N][3][C] totals; the reduction pass is skipped. */
int64_t gs_inorm_backward_scratch_floats(int32_t N, int32_t D, int32_t H, int32_t W, int32_t C);
/* gs_inorm_act_backward always leaves the per-image totals [N][3][C] (sum ghat, sum ghat*yhat, sum yhat) behind the
 * partial slots, at scratch + N * slots * 3 * C (slots = pre_slots, or the chunk count of its own reduction pass =
 * gs_inorm_backward_scratch_floats / (3 N C) - 1). An executor that passes bias_grad = NULL there can add the bias
 * gradients of ALL its norm layers with one launch afterwards: item i does db[c] += sum over images (in order) of
 * -rstd * S2 * S3 / hw. */
#define GS_NORM_DB_MAX 32
typedef struct gs_norm_db_item {
  const float* sums;        /* [N][3][C] totals left by gs_inorm_act_backward */
  const float* mean_rstd;   /* [N][2][C] */
  float* db;                /* [C], accumulated into */
  int32_t N, C;
  float inv_hw;             /* 1 / (D*H*W) */
  int32_t pad_;
} gs_norm_db_item;
int gs_norm_bias_grads(const gs_norm_db_item* items, int32_t count, void* stream);

/* Generalised form for skip-connection graphs (nn/generators/unet/unet2d.py:110-157): the normalised tensor is read
 * through up to two activations (LeakyReLU by the next down-conv, ReLU by the up-conv on the skip half of
 * torch.cat([x, y], 1)), outputs / gradient inputs are channel slices of wider concat buffers, and nn.Dropout(p) may sit
 * between the norm and the consumer's activation (mask = counter-based hash of (seed, image, element)).
 *   forward : v = drop(norm(y));  x1 = act1(v);  x2 = act2(v) (x2 optional)
 *   backward: ghat = mask/(1-p) * (g1*act1'(yhat) + g2*act2'(yhat)) (g2 optional), dy as in gs_inorm_act_backward.
 * mean_rstd == NULL: no norm, `y` holds a sign-preserving activation output. */
typedef struct gs_norm_ex_desc {
  int32_t N, H, W, C;
  int32_t act1, act2;          /* GS_ACT_* */
  float   slope;
  int32_t x1_cs, x1_co, x2_cs, x2_co;      /* forward outputs: channel stride / offset (elements) */
  int32_t g1_cs, g1_co, g2_cs, g2_co;      /* backward gradient inputs */
  float   drop_p;              /* 0 = no dropout */
  uint32_t seed_lo, seed_hi;   /* host part of the 64-bit mask seed */
  const uint32_t* seed_dev;    /* optional DEVICE part (2 words, lo/hi), added to the host part: lets a captured hipGraph
                                * of the step draw a fresh nn.Dropout mask per replay (unet2d.py:146-147) */
} gs_norm_ex_desc;
int gs_norm_act_forward_ex(const gs_norm_ex_desc* d, const void* y, const float* mean_rstd, void* x1, void* x2,
                           void* stream);
int gs_norm_act_backward_ex(const gs_norm_ex_desc* d, const void* g1, const void* g2, const void* y,
                            const float* mean_rstd, void* dy, float* scratch, float* bias_grad, void* stream);
int64_t gs_norm_backward_ex_scratch_floats(const gs_norm_ex_desc* d);

/* ---- V-Net elementwise family (nn/generators/vnet/vnet3d.py:155-267; memcnn.AdditiveCoupling via nn/invertible.py) ----
 * InstanceNorm3d(affine=False) -> [+ residual] -> nn.PReLU(C) -> [+ residual] on channel slices, and its backward incl.
 * the gradient of the learnable slope:
 *   forward : u = norm(y) (mean_rstd == NULL: u = y);  res_mode 1: u += res;  v = u > 0 ? u : slope[c]*u (slope == NULL:
 *             v = u);  res_mode 2: v += res;  res_mode 3: v = res - v (the inverse of an additive coupling, x = y - F(.),
 *             memcnn AdditiveCoupling.inverse through nn/invertible.py:21-24);  out = v
 *   backward: gt = g (+ g2), negated for res_mode 3;  gu = gt*(u > 0 ? 1 : slope[c]);  dslope[c] += sum gt*min(u, 0);  gres = gu (optional);
 *             dy = rstd*(gu - mean gu - yhat*mean(gu*yhat)) (no norm: dy = gu);  bias_grad[c] += sum over pixels of dy
 * res_mod > 0: channel c reads residual channel c % res_mod (InputBlock's x.repeat, vnet3d.py:162-167). */
typedef struct gs_pnorm_desc {
  int64_t pixels;                           /* D*H*W of one image */
  int32_t N, C;                             /* C multiple of 8 */
  int32_t y_cs, y_co;                       /* channel stride / offset (elements) of every operand view */
  int32_t res_mode, res_cs, res_co, res_mod;
  int32_t out_cs, out_co;
  int32_t g_cs, g_co, g2_cs, g2_co;
  int32_t dy_cs, dy_co, gres_cs, gres_co;
} gs_pnorm_desc;
int gs_pnorm_forward(const gs_pnorm_desc* d, const void* y, const float* mean_rstd, const void* res, const float* slope,
                     void* out, void* stream);
int gs_pnorm_backward(const gs_pnorm_desc* d, const void* g, const void* g2, const void* y, const float* mean_rstd,
                      const void* res, const float* slope, void* dy, void* gres, float* dslope, float* bias_grad,
                      float* scratch, void* stream);
int64_t gs_pnorm_backward_scratch_floats(const gs_pnorm_desc* d);
/* dst[view] = src[view] (accumulate == 0) or dst[view] += src[view]: gradient joins (out + down, torch.cat skip halves,
 * vnet3d.py:198-200,239-243) */
int gs_add_views(void* dst, int32_t dst_cs, int32_t dst_co, const void* src, int32_t src_cs, int32_t src_co,
                 int64_t pixels, int32_t C, int32_t accumulate, void* stream);
/* adjoint of x.repeat(1, C/Cin, 1, 1, 1): g_img[n][c0][pix] += sum over c = c0 (mod Cin) of g[pix][c] */
int gs_repeat_backward(const void* g, int32_t g_cs, int32_t g_co, float* g_img, int32_t N, int32_t Cin, int32_t C,
                       int64_t pixels, void* stream);

/* ---- network boundary: NCHW fp32 images <-> NHWC bf16 activations ------------------------------- */
/* x NCHW fp32 [N,C,H,W] -> act [N,H,W,Cp]  (set_input, cyclegan.py:84-90) */
int gs_image_to_act(const float* img, void* act, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                    void* stream);
/* torch.cat([a, b], dim=1) -> act in one pass (the conditional discriminator's input, pix2pix.py:70,80), and its gradient into
 * the two images' own gradient tensors (ga / gb may be NULL: that image needs none) */
int gs_image_pair_to_act(const float* a, int32_t Ca, const float* b, int32_t Cb, void* act, int32_t N, int32_t H, int32_t W,
                         int32_t Cp, void* stream);
int gs_image_pair_to_act_backward(const void* g, float* ga, int32_t Ca, float* gb, int32_t Cb, int32_t N, int32_t H, int32_t W,
                                  int32_t Cp, void* stream);
/* act -> NCHW fp32, optional tanh (resnet2d.py:65) */
int gs_act_to_image(const void* act, float* img, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                    int32_t act_kind, void* stream);
/* gradient of gs_act_to_image: g_img NCHW fp32, out = forward output image (for tanh') -> g_act */
int gs_act_to_image_backward(const float* g_img, const float* out_img, void* g_act, int32_t N, int32_t C,
                             int32_t H, int32_t W, int32_t Cp, int32_t act_kind, void* stream);
/* gradient of gs_image_to_act composed with the first conv's padding fold: g_pad act (padded by fold)
 * -> NC(D)HW fp32 image gradient; accumulate != 0 adds into g_img. The other three boundary functions are
 * layout-only: a volume goes through them with H := D*H. */
int gs_image_to_act_backward(const void* g_pad, float* g_img, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                             int32_t Cp, int32_t fold, int32_t fold_mode, int32_t accumulate, void* stream);

/* ---- "W-fold" of the k7 boundary convolutions (resnet2d.py:24-25,64-65; resnet3d.py:24-25,64) --------------------
 * A conv with 1-3 input (stem) or output (last layer) channels wastes the 16-wide matrix tile, so the taps of the W
 * axis are moved into the channel axis and the conv itself runs with a k x [k x] 1 kernel through gs_gconv_forward /
 * gs_wgrad; these are the boundary transforms and their adjoints. `rows` = D*H of a volume (or H of an image).
 *   unfold   : act[n, r, j][dw*C + c] = img[n][c][r][B(j + dw - p)], channels >= k*C zero (fused NCHW fp32 -> NHWC bf16)
 *   shift-add: img[n][co][r][j] = act(bias[co] + sum_dw z[n, r, j + dw][dw*Co + co]),  z is W + k - 1 wide */
int gs_image_unfold(const float* img, void* act, int32_t N, int32_t C, int64_t rows, int32_t W, int32_t Qp, int32_t k,
                    int32_t p, int32_t border, void* stream);
/* adjoint of gs_image_unfold composed with the (depth, row) padding fold of the stem conv: g is the data gradient of the
 * transformed conv on the domain padded by `fold` along depth (D > 1) and rows, W unpadded */
int gs_image_unfold_backward(const void* g, float* g_img, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                             int32_t Qp, int32_t k, int32_t p, int32_t fold, int32_t border, int32_t accumulate,
                             void* stream);
int gs_shiftadd_to_image(const void* z, const float* bias, float* img, int32_t N, int32_t Co, int64_t rows, int32_t W,
                         int32_t Pp, int32_t k, int32_t act_kind, void* stream);
/* gz[n, r, j'][dw*Co + co] = g_img[n][co][r][j' - dw] * act'(out) (zero outside [0, W)), channels >= k*Co zero */
int gs_shiftadd_to_image_backward(const float* g_img, const float* out_img, void* gz, int32_t N, int32_t Co,
                                  int64_t rows, int32_t W, int32_t Pp, int32_t k, int32_t act_kind, void* stream);

/* Partial InstanceNorm statistics of a channel slice [co, co + C) of an activation tensor with channel stride cs — for a
 * norm whose input is not a conv output (the pre-norm of Piresnet3D's coupling function, piresnet3d.py:104-108):
 * partial [N][slots][2][C] with slots = gs_slice_stats_slots(pixels), finalised by gs_inorm_finalize. */
int32_t gs_slice_stats_slots(int64_t pixels);
int gs_slice_stats(const void* x, int32_t N, int64_t pixels, int32_t cs, int32_t co, int32_t C, float* partial,
                   void* stream);

/* ---- device-side image preprocessing --------------------------------------------- */
/* What ganslate/data/utils/transforms.py:9-61 composes on the host from torchvision / PIL for the image-folder datasets
 * (unpaired_image_dataset.py:31-62, paired_image_dataset.py): Resize(load_size, Image.BICUBIC) / scale_width / random_zoom, RandomCrop(final_size),
 * RandomHorizontalFlip, ToTensor, Normalize(0.5, 0.5) — here on decoded 8-bit HWC images (C = 1 or 3) in device memory.
 * The resize is Pillow's two-pass 8-bit resampler bit for bit (src/libImaging/Resample.c): the caller supplies, per axis,
 * bounds[out][2] = (first input index, count) and kk[out][ksize] = 22-bit fixed-point coefficients
 * (precompute_coeffs + normalize_coeffs_8bpc; identity tables when an axis keeps its size).
 * Pass 1, horizontal: out[y][xx][c] = clip8((2^21 + sum_k in[y][xmin+k][c] * kk[xx][k]) >> 22), all in_h rows. */
int gs_u8_resample_h(const void* in, void* out, int32_t in_h, int32_t in_w, int32_t out_w, int32_t C,
                     const int32_t* bounds, const int32_t* kk, int32_t ksize, void* stream);
/* Pass 2 as a plain 8-bit image (out_h x w): the intermediate of two consecutive resizes — `scale_width` / `resize` followed by
 * `random_zoom` (transforms.py:22-37,127-137,163-169). */
int gs_u8_resample_v(const void* tmp, void* out, int32_t tmp_h, int32_t w, int32_t out_h, int32_t C, const int32_t* bounds,
                     const int32_t* kk, int32_t ksize, void* stream);
/* Pass 2, vertical, only for the crop window [top, top+fh) x [left, left+fw) of the out_h x tmp_w resized image, then flip,
 * x/255 and (x-0.5)/0.5 in torchvision's fp32 order: out[c][i][j] (planes of fh*fw floats — a slice of the NCHW batch). */
int gs_u8_resample_v_crop_normalize(const void* tmp, float* out, int32_t tmp_h, int32_t tmp_w, int32_t out_h, int32_t C,
                                    const int32_t* bounds, const int32_t* kk, int32_t ksize, int32_t top, int32_t left,
                                    int32_t fh, int32_t fw, int32_t flip, void* stream);

/* ---- device-side 3-D training patches ------------------------------------------- */
/* What the 3-D datasets' workers do per sample on the host (projects/brats_mri_sequence_translation/datasets/
 * train_dataset.py:83-86 -> ganslate/data/utils/normalization.py:18-30): out = z_score_normalize(volume[z:z+d, y:y+h,
 * x:x+w], scale_to_range=(lo, hi)) — t = (v - mean) / std with the patch's own mean and unbiased std, then
 * (hi - lo) * (t - min t) / (max t - min t) + lo when rescale != 0 — on a dense D x H x W volume resident in device
 * memory (dtype GS_VOL_F32 or GS_VOL_I16). start / size: host arrays {z, y, x} / {d, h, w}; out: d*h*w floats;
 * scratch: gs_patch_zscore_ws_floats() floats, 8-byte aligned, private to the launch. A constant patch gives NaN, as in
 * the reference. The start coordinates come from data/utils/stochastic_focal_patching.py (host RNG). */
enum { GS_VOL_F32 = 0, GS_VOL_I16 = 1 };
int64_t gs_patch_zscore_ws_floats(void);
int gs_patch_zscore(const void* vol, int32_t dtype, int32_t D, int32_t H, int32_t W, const int32_t* start,
                    const int32_t* size, int32_t rescale, float lo, float hi, float* out, float* scratch, void* stream);

/* ---- losses (fp32, on the boundary images / discriminator maps) --------------------------------- */
/* loss[0] = mean((x-target)^2); if grad != NULL: grad = grad_scale * 2*(x-target)/n
 * (nn.MSELoss vs expanded constant, adversarial_loss.py:28-29,60-62) */
int gs_mse_const(const float* x, int64_t n, float target, float* loss, float* grad, const float* grad_scale,
                 void* stream);
/* Every objective of AdversarialLoss.calculate_loss (adversarial_loss.py:52-73) on one discriminator map of n floats.
 * mode GS_ADV_LSGAN      mean((x-label)^2)                                    (= gs_mse_const)
 *      GS_ADV_VANILLA    nn.BCEWithLogitsLoss vs the expanded label            (:31-32,60-62)
 *      GS_ADV_WGANGP     -mean(x) if target_is_real else +mean(x)             (:63-67)
 *      GS_ADV_NONSAT     per-sample mean of softplus(-x) / softplus(x): loss and grad_scale hold `rows` floats, rows = the
 *                        batch (:68-73; the reference's branch dies on an unimported F — this is what it spells out)
 * loss and/or grad (n floats, = grad_scale * d loss / d x) are written; either may be NULL. */
enum { GS_ADV_LSGAN = 0, GS_ADV_VANILLA = 1, GS_ADV_WGANGP = 2, GS_ADV_NONSAT = 3 };
int gs_adv_loss(const float* x, int64_t n, int32_t rows, int32_t mode, int32_t target_is_real, float label,
                float* loss, float* grad, const float* grad_scale, void* stream);
/* loss[0] = mean(|a-b|); if grad_a != NULL: grad_a = grad_scale * sign(a-b)/n
 * (nn.L1Loss, cyclegan_losses.py:64,75,97-101) */
int gs_l1(const float* a, const float* b, int64_t n, float* loss, float* grad_a, const float* grad_scale,
          void* stream);
/* out[0] = mean(x)  (train_metrics.py:27-33) */
int gs_mean(const float* x, int64_t n, float* out, void* stream);
/* out[r] = c[r] + sum_k m[r*K + k] * x[k][0], r < R <= 8, k < K <= 16: the scalar algebra of a recipe's loss assembly
 * (lambda * (alpha * ssim + beta * l1), loss_real + loss_fake, the sum of the G losses: cyclegan_losses.py:21-32,70-90,
 * cyclegan.py:150,182) as one launch. x = HOST array of K device pointers to fp32 scalars (a null entry counts as 0), m / c =
 * host arrays (c may be null). The backward of a combination is the same call with the transposed matrix over the rows'
 * upstream gradients. */
int gs_scalar_affine(const float* const* x, int32_t K, const float* m, const float* c, int32_t R, float* out, void* stream);
/* out = a + b over n floats (16-byte aligned): the join of the two gradients of a generated image that feeds a
 * discriminator and the other generator (cyclegan.py:131-141) */
int gs_sum2_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* SSIM distance of ganslate/nn/losses/utils/ssim.py:65-99 on (x+1)/2,(y+1)/2, window 11, sigma 1.5:
 * out[0] = mean(sqrt(relu(2 - S1 - S2))) over [N*C, H-10, W-10] */
int gs_ssim_distance(const float* x, const float* y, int32_t NC, int32_t H, int32_t W, float* out,
                     float* scratch, void* stream);
int64_t gs_ssim_scratch_floats(int32_t NC, int32_t H, int32_t W);
/* gradient of gs_ssim_distance w.r.t. y (the distance is symmetric: swap x and y for the other one), scaled by the
 * upstream scalar grad_scale[0] (device pointer, NULL = 1): the SSIM-weighted cycle loss, cyclegan_losses.py:78-90 */
int gs_ssim_distance_backward(const float* x, const float* y, int32_t NC, int32_t H, int32_t W,
                              const float* grad_scale, float* grad_y, float* scratch, void* stream);
int64_t gs_ssim_backward_scratch_floats(int32_t NC, int32_t H, int32_t W);

/* ---- PatchNCE + patch MLP of CUT (ganslate/nn/gans/unpaired/cut.py:229-294, ganslate/nn/losses/cut_losses.py:14-43) ----
 * For every feature level l: sampled patches xq[l], xk[l] are [batch*patches][channels[l]] fp32 (target = query, source =
 * key, row = image * patches + patch); FeaturePatchMLP level l = Linear(C_l, nc) - ReLU - Linear(nc, nc) - x/(||x||+1e-7);
 * PatchNCE: logits [q.k+ , q.k_j over the other patches of the same image, diagonal -> -10] / nce_T, cross-entropy vs 0,
 * keys detached. params / grads: one flat fp32 buffer, per level W1 [nc][C_l], b1 [nc], W2 [nc][nc], b2 [nc]
 * (gs_patchnce_param_floats). loss[l] = lambda_nce / (levels * batch * patches) * sum over rows of the row loss, so that
 * sum_l loss[l] is what CUT._calculate_nce_loss returns (cut.py:218-226).
 * forward leaves what backward needs in `work` (gs_patchnce_work_bytes, caller-owned, same buffer for both calls);
 * backward: dxq[l] = d(sum_l loss[l]) / d xq[l] * grad_scale[0], grads += parameter gradients * grad_scale[0]
 * (grad_scale: device scalar, NULL = 1). nc must be 256 and patches <= 256 (the reference's defaults, cut.py:16-20). */
#define GS_PATCHNCE_MAX_LEVELS 8
typedef struct gs_patchnce_desc {
  int32_t levels, batch, patches, nc;
  int32_t channels[GS_PATCHNCE_MAX_LEVELS];
  float nce_T, lambda_nce;
} gs_patchnce_desc;
int64_t gs_patchnce_param_floats(const gs_patchnce_desc* d);
int64_t gs_patchnce_work_bytes(const gs_patchnce_desc* d);
int gs_patchnce_forward(const gs_patchnce_desc* d, const float* const* xq, const float* const* xk, const float* params,
                        void* work, float* loss, void* stream);
int gs_patchnce_backward(const gs_patchnce_desc* d, const float* const* xq, float* const* dxq, const float* params,
                         float* grads, void* work, const float* grad_scale, void* stream);

/* ---- SelfAttentionBlock (ganslate/nn/attention.py:12-47; selfattention_patchgan3d.py:58,73, selfattention_vnet3d.py:97-104) ----
 * x, out, dout, dx: NDHWC bf16 [B][N][C] (N = D*H*W voxels of the map, C a multiple of 8). q / k = 1x1x1 convs to C/8 channels,
 * v to C channels; A = softmax over keys of q_i . k_j; out = gamma * (A v) + x. Parameters in torch layout, fp32:
 * wq, wk [C/8][C], bq, bk [C/8], wv [C][C], bv [C], gamma [1] (device pointers). forward leaves q / k / v, the bf16 attention
 * matrix [B][N][N] and A v in `work` (gs_attn_work_bytes, caller-owned, the same buffer for both calls); backward writes
 * dx = d loss / d x and ADDS the parameter gradients into the tensors of `grads` (NULL members / NULL grads: skipped). */
typedef struct gs_attn_desc { int32_t B, N, C; } gs_attn_desc;
typedef struct gs_attn_params { float *gamma, *wq, *bq, *wk, *bk, *wv, *bv; } gs_attn_params;
int64_t gs_attn_work_bytes(const gs_attn_desc* d);
/* ... and for a forward pass whose state is never handed to gs_attn_backward (inference, attention.py:26-47 under no_grad):
 * without the backward's buffers */
int64_t gs_attn_forward_work_bytes(const gs_attn_desc* d);
int gs_attn_forward(const gs_attn_desc* d, const void* x, const gs_attn_params* params, void* out, void* work, void* stream);
int gs_attn_backward(const gs_attn_desc* d, const void* x, const void* dout, const gs_attn_params* params,
                     const gs_attn_params* grads, void* work, void* dx, void* stream);

/* ---- optimiser (torch.optim.Adam betas=(0.5,0.999) eps=1e-8, cyclegan.py:81-82) ------------------ */
/* hyper (host pointer to 6 floats): lr, beta1, beta2, eps, bias_correction1, sqrt(bias_correction2).
 * grad_scale multiplies the gradient (1/world_size after an all-reduce SUM). zero_grad != 0 clears g. */
int gs_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_host,
                 float grad_scale, int32_t zero_grad, void* stream);
/* Same update with `hyper` (same 6 floats) in DEVICE memory, so a captured hipGraph of the step can be replayed while
 * the host refreshes the learning rate and bias corrections between replays. */
int gs_adam_step_dev(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                     float grad_scale, int32_t zero_grad, void* stream);
/* gs_adam_step_dev that also refreshes the bf16 weight packs where 8 consecutive pack elements are 8 consecutive, 8-aligned
 * master elements (row-major packs: a conv's forward pack, a transposed conv's data-gradient pack; nn/native/net.py builds
 * the tables): inv_x[i] = pack group (of 8 elements) holding master elements 8 i .. 8 i + 7, or -1; tables of ceil(n / 8)
 * int32. Either table / pack pair may be NULL. The transposed packs keep gs_repack_bf16_tiled_groups. Buffers 16-byte
 * aligned. (torch.optim.Adam.step + the weight cast a bf16 autocast run would do per use, pix2pix.py:61-66) */
int gs_adam_step_dev_packs(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_dev, float grad_scale,
                           int32_t zero_grad, const int32_t* inv_f, void* fpack, const int32_t* inv_d, void* dpack,
                           void* stream);
/* gs_adam_step_dev_packs over several ranges [start, end) of the flat buffers in ONE launch (ranges_dev: device int64 [n_ranges][2],
 * elements, starts multiples of 8; max_len = the longest range; inv_f / inv_d index the WHOLE buffers): what gs_wgrad_adam leaves of
 * a network — biases and small layers between the layers it updated itself */
int gs_adam_step_dev_packs_ranges(float* p, float* g, float* m, float* v, const int64_t* ranges_dev, int32_t n_ranges,
                                  int64_t max_len, const float* hyper_dev, float grad_scale, int32_t zero_grad,
                                  const int32_t* inv_f, void* fpack, const int32_t* inv_d, void* dpack, void* stream);
/* ---- the PatchGAN's last layer: Conv2d(8 ndf, 1, k4, s1, p1) (patchgan2d.py:62) — one output channel -------------------------
 * A dot product per pixel: on the vector ALUs (v_dot2c_f32_bf16) with the filter in registers and a sliding 4 x 4 window of
 * input pixels, instead of an eighth of an MFMA tile behind a 16-fold im2col gather (csrc/cout1.hip). Same descriptors,
 * packs (row 0 is the filter; the other rows of the 8-row pack are zero) and results as gs_gconv_forward / gs_wgrad_ws for
 * this layer: out channel 0 = conv + bias, channels 1..7 = 0; dw row 0 += the gradient (rows 1..7 belong to channels whose
 * output gradient is zero). _eligible: 2-D, 16 taps forming a 4 x 4 grid, stride 1, zero border, Co = 8, Ci <= 512, no
 * statistics / accumulate. tw may be NULL. Workspace of the weight gradient: gs_wgrad_cout1_ws_floats (-1: not eligible);
 * partial sums per workgroup are added in a fixed order. */
int gs_conv_cout1_eligible(const gs_gconv_desc* d);
int gs_conv_cout1_forward(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                          const gs_twin* tw, void* stream);
int gs_wgrad_cout1_eligible(const gs_wgrad_desc* d);
int64_t gs_wgrad_cout1_ws_floats(const gs_wgrad_desc* d);
int gs_wgrad_cout1_ws(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, float* ws, int64_t ws_floats,
                      const gs_twin* tw, void* stream);

/* ---- feature taps of CUT's PatchNCE loss (ganslate/nn/gans/unpaired/cut.py:229-312; FeaturePatchMLP.forward :262-277 reads
 * `feat.permute(0, 2, 3, 1).flatten(1, 2)[:, patch_id, :]` per level) ------------------------------------------------------------
 * ids_dev: int64 device array of P DISTINCT flat pixel indices (the head of a torch.randperm), shared by the n images.
 * gs_tap_gather: out[n][p][ch] = src[n][ids[p]][ch] as fp32, ch < c; src = n images of `pixels` NHWC bf16 pixels, cs channels.
 * gs_tap_scatter_add: the backward of it into a gradient buffer on a domain padded by f0: with (y, x) = divmod(ids[p], W),
 *   dst[n][(y + f0) * Wp + x + f0][ch] += g[n][p][ch] (bf16 storage, fp32 add); `pixels` = pixels per image of dst.
 * gs_tap_rows_sum: db[ch] += sum over rows of g[rows][c] in a fixed order (bias gradient seen by a tap of a raw conv output).
 * gs_zero_bytes: zero a 16-byte aligned buffer (the start of a gradient that only taps feed).
 * gs_image_tap_gather / _scatter: nce layer 0 is the ReflectionPad2d(pad) output of the fp32 NCHW image (resnet2d.py:24):
 *   out[n][p][ch] = x[n][ch][r(yp - pad)][r(xp - pad)], (yp, xp) = divmod(ids[p], W + 2 pad), r = reflection; the scatter
 *   zeroes gx [N][C][H][W] and adds g through the same map, colliding samples (reflected onto one pixel) in sample order:
 *   deterministic, at the price of an O(P^2) scan of the ids in LDS — P <= 16384 samples per image (the reference draws
 *   256, cut.py:46), larger P is refused with an error. */
int gs_tap_gather(const void* src, int32_t n, int64_t pixels, int32_t cs, const int64_t* ids_dev, int32_t P, int32_t c, float* out,
                  void* stream);
int gs_tap_scatter_add(void* dst, int32_t n, int64_t pixels, int32_t cs, const int64_t* ids_dev, int32_t P, int32_t c, int32_t W,
                       int32_t Wp, int32_t f0, const float* g, void* stream);
int gs_tap_rows_sum(const float* g, int64_t rows, int32_t c, float* db, void* stream);
int gs_zero_bytes(void* p, int64_t bytes, void* stream);
int gs_image_tap_gather(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t pad, const int64_t* ids_dev, int32_t P,
                        float* out, void* stream);
int gs_image_tap_scatter(const float* g, int32_t N, int32_t C, int32_t H, int32_t W, int32_t pad, const int64_t* ids_dev,
                         int32_t P, float* gx, void* stream);
/* FastCUT's flip-equivariance coin (cut.py:146-152: `real_A.flip(-1)` on a host coin): out = flag_dev[0] ? x.flip(-1) : x over
 * [rows][W] fp32, the flag in device memory so that a captured step replays either way. Not in place. */
int gs_flip_w_if(const float* x, float* out, int64_t rows, int32_t W, const int32_t* flag_dev, void* stream);

/* ImagePool.query (ganslate/data/utils/image_pool.py:31-60) with the host's coin flips uploaded as code_dev[B]:
 * < 0 pass image b through; slot: store image b in `slot`, return it (pool filling); slot | 0x40000000: return the
 * image stored in `slot`, store image b there. Images of a batch are handled in order (same-slot draws chain like the
 * reference's loop). pool = [slots][image_bytes], images/out = [B][image_bytes], image_bytes % 16 == 0. */
int gs_pool_query(void* pool, const void* images, void* out, const int32_t* code_dev, int32_t B,
                  int64_t image_bytes, void* stream);
/* pack[e] = bf16(master[index[e]]) (index < 0 -> 0): refresh the bf16 forward/dgrad weight packs */
int gs_repack_bf16(const float* master, const int32_t* index, void* pack, int64_t n, void* stream);

/* Same refresh for one [rows][kp] pack segment whose master indices run along the rows (transposed packs): 64 x 64
 * tiles through LDS so that both the master reads and the pack writes are contiguous. Results are identical. */
int gs_repack_bf16_tiled(const float* master, const int32_t* index, void* pack, int32_t rows, int32_t kp, void* stream);
/* Group-indexed forms of the two refreshes (round 3): one base index per 8 pack elements whose sources are 8 consecutive
 * master elements. Same call sites as gs_repack_bf16 (optimizer.step() -> weights the next forward reads,
 * cyclegan.py:107,123); an eighth of the index traffic; two launches per pack of a network.
 * gs_repack_bf16_groups, along k over the WHOLE pack: pack[8 g + j] = master[gindex[g] + j]; gindex[g] = -1 -> zeros,
 *   -2 -> the group's eight own entries of the element-wise table `index` (null if no group is marked), -3 -> not written
 *   (the group belongs to a transposed segment); n8 groups.
 * gs_repack_bf16_tiled_groups, along the rows of ALL transposed segments of the pack: seg_dev[5 i ..] = {pack offset in
 *   elements, gindex offset, rows (multiple of 8), kp (multiple of 64), first tile}, tiles = sum of ceil(rows / 64) * kp / 64;
 *   pack[off + (8 G + j) * kp + k] = master[gindex[goff + G * kp + k] + j], negative -> zeros.
 * master 16-byte aligned, pack 16-byte (8-byte for the segments) aligned. */
int gs_repack_bf16_groups(const float* master, const int32_t* gindex, const int32_t* index, void* pack, int64_t n8,
                          void* stream);
int gs_repack_bf16_tiled_groups(const float* master, const int32_t* gindex, void* pack, const int64_t* seg_dev,
                                int32_t nseg, int64_t tiles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GANSLATE_HIP_H */
