/*
 * pwg_kernels.h -- C ABI of libpwgkernels.so, the MI355X (gfx950) kernel library
 * behind the ParallelWaveGAN-compatible Python surface in parallelwavegan_amd/.
 *
 * The reference (kan-bayashi/ParallelWaveGAN) has no native code: every FLOP of
 * its hot path is an ATen op called from Python.  Each entry
 * point below therefore replaces an ATen *call site* in the reference; the
 * file:line of that call site (relative to /root/reference/) is cited on every
 * declaration.  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - plain C: raw device pointers (fp32, contiguous NCW), ints, floats; no
 *     torch types.  `stream` is a hipStream_t passed as void*.
 *   - the library allocates nothing; every buffer incl. packed weights and
 *     workspaces is owned by the caller.
 *   - every launch goes to `stream`; no implicit device synchronisation.
 *   - return value: PWG_OK (0) or a negative pwg_status; pwg_last_error()
 *     returns a thread-local human-readable message for the last failure.
 */
#ifndef PWG_KERNELS_H_
#define PWG_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pwg_status {
  PWG_OK = 0,
  PWG_ERR_BAD_SHAPE = -1,   /* inconsistent sizes in a descriptor            */
  PWG_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for     */
  PWG_ERR_LAUNCH = -3,      /* hipLaunchKernel / hipGetLastError failed        */
  PWG_ERR_NULL = -4,        /* required pointer is NULL                        */
  PWG_ERR_WORKSPACE = -5    /* workspace too small                             */
} pwg_status;

const char* pwg_last_error(void);
/* ABI version: bumped on any signature change. */
int pwg_abi_version(void);
/* Returns 950 (the only ISA this library is built for). */
int pwg_target_arch(void);

/* ------------------------------------------------------------------------- */
/* Per-launch timing (measurement support).  While enabled, every */
/* kernel launched through this ABI is bracketed by hipEvents recorded on the   */
/* stream it is launched on; totals are kept per kernel family together with     */
/* the ALGORITHMIC flops/bytes of the launches (not hardware counters).          */
/* pwg_prof_get/_num_kernels synchronise the recorded events.                     */
/* ------------------------------------------------------------------------- */
int pwg_prof_enable(int on);
int pwg_prof_reset(void);
int pwg_prof_num_kernels(void);
int pwg_prof_get(int32_t idx, char* name, size_t name_cap, double* total_ms, int64_t* launches,
                 double* flops, double* bytes);

/* Concurrency hint for the tile / split heuristics of the convolution kernels: the fraction of the chip ONE launch
 * should aim to fill (1.0 = it runs alone, the default; 0.5 = about two launches run concurrently, e.g. the parallel
 * sub-discriminator branches of a captured training step, so fewer but more efficient tiles win).  Process-wide;
 * values outside (0, 1] are ignored; returns the previous value.  Measured on the HiFi-GAN V1 step: +2.7 %.      */
float pwg_set_concurrency_hint(float fill_scale);

/* Debugging aid (tests): while on, every CU's LDS is filled with NaN bit patterns before each MFMA    */
/* kernel launched through this ABI, so that a result depending on LDS nobody wrote (0 * stale tile      */
/* element in a contraction) turns non-finite deterministically.  Same as the environment variable       */
/* PWG_POISON_LDS=1, switchable at run time; returns the previous setting.                                */
int pwg_debug_poison_lds(int on);

/* ------------------------------------------------------------------------- */
/* Activations / padding selectors                                            */
/* ------------------------------------------------------------------------- */
enum { PWG_ACT_NONE = 0, PWG_ACT_LEAKY_RELU = 1, PWG_ACT_TANH = 2, PWG_ACT_RELU = 3 };
enum { PWG_PAD_ZERO = 0, PWG_PAD_REFLECT = 1, PWG_PAD_REPLICATE = 2 };

/* ------------------------------------------------------------------------- */
/* 1-D convolution family                                                     */
/*                                                                            */
/* One descriptor covers every convolution on the hot path:                   */
/*   - torch.nn.Conv1d (dilated / strided / grouped), e.g.                    */
/*       layers/residual_block.py:190-196,213-220  (HiFi-GAN MRF)             */
/*       layers/residual_block.py:78-86             (PWG dilated gate conv)   */
/*       models/hifigan.py:75-81,143-149            (input / output conv)     */
/*       models/hifigan.py:516-568                  (MSD grouped strided)     */
/*       layers/residual_stack.py:47-55             (MelGAN stack, reflect)   */
/*   - torch.nn.ConvTranspose1d, computed polyphase (transposed = 1):         */
/*       models/hifigan.py:99-107, models/melgan.py:93-101                    */
/*   - torch.nn.Conv2d with (k,1) kernels on the period-folded view           */
/*       (width = period): models/hifigan.py:314-341                          */
/*                                                                            */
/* Fused around the contraction (all optional):                               */
/*   y = post_act( (conv(pre_act(pad(x)), w) + bias + add1 + add2) * out_mul  */
/*                  / out_div )                                               */
/* which is how `x = xt + x` (residual_block.py:257), the MRF `cs += ...;     */
/* c = cs / num_blocks` (hifigan.py:186-190) and the final Tanh               */
/* (hifigan.py:150) disappear into the producing kernel.                      */
/* ------------------------------------------------------------------------- */
typedef struct pwg_conv1d_desc {
  int32_t batch;
  int32_t c_in;       /* total input channels                                 */
  int32_t c_out;      /* total output channels                                */
  int32_t t_in;       /* input length  (rows H for the (k,1) Conv2d case)     */
  int32_t t_out;      /* output length (rows H_out)                           */
  int32_t width;      /* 1 for Conv1d; the period p for (k,1) Conv2d          */
  int32_t kernel;     /* taps                                                 */
  int32_t stride;
  int32_t dilation;
  int32_t pad_left;   /* implicit padding on the left (zeros/reflect/...)     */
  int32_t groups;
  int32_t transposed; /* 0: Conv1d semantics, 1: ConvTranspose1d semantics    */
                      /*    (stride = upsampling factor, pad_left = `padding`)*/
  int32_t pad_mode;   /* PWG_PAD_*  (reflect/replicate only for width==1)     */
  int32_t pre_act;    /* PWG_ACT_NONE | LEAKY_RELU | RELU applied to x        */
  float pre_slope;
  int32_t post_act;   /* PWG_ACT_* applied to the result                      */
  float post_slope;
  float out_mul;      /* 1.0f = off                                           */
  float out_div;      /* 1.0f = off  (true IEEE division, matches `cs / 3`)   */
} pwg_conv1d_desc;

/* Number of floats of the packed weight image for `d` (forward direction). */
size_t pwg_conv1d_packed_weight_floats(const pwg_conv1d_desc* d);

/* Re-layout a torch-format weight into the kernel's [group][tap][ci][M] image.
 *   transposed == 0: w is (c_out, c_in/groups, kernel)   [torch Conv1d]
 *   transposed == 1: w is (c_in, c_out/groups, kernel)   [torch ConvTranspose1d]
 * `scale` (optional, may be NULL) holds one multiplier per dim-0 slice of w,
 * i.e. g/||v|| of old-style weight_norm (see pwg_weight_norm_scale), so that
 * weight_norm + pack is one pass.                                             */
int pwg_conv1d_pack_weight(const pwg_conv1d_desc* d, const float* w, const float* scale,
                           float* w_packed, void* stream);

/* y = fused conv forward (see above).  bias/add1/add2 may be NULL; y must not alias any input.
 * x: (batch, c_in, t_in*width)  y/add1/add2: (batch, c_out, t_out*width).
 * workspace (optional, may be NULL/0): launches that cannot fill the chip but have a long
 * reduction (e.g. 1024 channels over 9..110 columns per item) are run as 2/4/8 reduction
 * slices of a bigger tile; each slice writes a y-shaped slab of `workspace` and a second kernel
 * sums the slabs in slice order and applies the fused terms (deterministic).  Query the size
 * (floats, 0 = not needed) first; without a workspace the launch runs unsplit.              */
size_t pwg_conv1d_forward_workspace_floats(const pwg_conv1d_desc* d);
int pwg_conv1d_forward(const pwg_conv1d_desc* d, const float* x, const float* w_packed,
                       const float* bias, const float* add1, const float* add2, float* y,
                       float* workspace, size_t workspace_floats, void* stream);

/* ---- backward (training; replaces ATen convolution_backward at the same call sites) ---- */
/* Weight image for the data-gradient direction (the dual convolution).          */
size_t pwg_conv1d_packed_weight_bwd_floats(const pwg_conv1d_desc* d);
int pwg_conv1d_pack_weight_bwd(const pwg_conv1d_desc* d, const float* w, const float* scale,
                               float* w_packed_bwd, void* stream);
/* ---- weight bank (round 4): the weight preparation of a whole model in two launches ----
 * Replaces the per-layer sequence pwg_weight_norm_scale + pwg_conv1d_pack_weight + pwg_conv1d_pack_weight_bwd
 * (i.e. the reference's per-layer weight-norm hook, torch/nn/utils/weight_norm.py `_weight_norm`, evaluated before
 * every F.conv1d call of the layers/ and models/ modules) by ONE row-scale launch and ONE packing launch over a device table.
 * pwg_weight_bank_build fills a HOST table (pwg_weight_bank_table_bytes(n) bytes) and `info[8]`; the caller copies
 * the table to the device once and passes that copy to pwg_weight_bank_prepare, whose launches have constant
 * arguments (hipGraph-capturable).  Results are bit-identical to the per-layer entry points.
 * item: w = weight / weight_v (torch layout), g = weight_g or NULL, scale = n0 floats (iff g), fwd / bwd = images of
 * pwg_conv1d_packed_weight_floats / _bwd_floats floats (NULL: not built), desc = any descriptor of the layer.   */
typedef struct pwg_bank_item {
  const float* w;
  const float* g;
  float* scale;
  float* fwd;
  float* bwd;
  pwg_conv1d_desc desc;
} pwg_bank_item;
size_t pwg_weight_bank_table_bytes(int32_t n_items);
int pwg_weight_bank_build(const pwg_bank_item* items, int32_t n_items, void* table_host, size_t table_bytes,
                          int32_t* info);
int pwg_weight_bank_prepare(const void* table_dev, const int32_t* info, int32_t with_bwd, void* stream);

/* dx = d(pre_act)/dx(x) * conv_data_grad(dy) + accum.  `d` is the FORWARD descriptor
 * (its post_act/out_mul/out_div are NOT differentiated here: the caller passes the gradient
 * w.r.t. the pre-post_act, pre-scale result).  x: forward input (may be NULL when
 * pre_act == NONE); accum: optional tensor added to the result (gradient accumulation);
 * it must NOT alias dx (the kernels treat outputs as restrict).  workspace: as for
 * pwg_conv1d_forward (split reduction), sized by the query below.                    */
size_t pwg_conv1d_backward_data_workspace_floats(const pwg_conv1d_desc* d);
int pwg_conv1d_backward_data(const pwg_conv1d_desc* d, const float* dy, const float* w_packed_bwd,
                             const float* x, const float* accum, float* dx, float* workspace,
                             size_t workspace_floats, void* stream);
/* dw (torch layout, same shape as the forward weight) = sum_{b,t} dy * pre_act(x) taps;
 * db[c] = sum dy.  The (batch, time) reduction is cut into slices that each write a private
 * slab of `workspace`; a second kernel sums the slabs (deterministic, no atomics).  For plain
 * convolutions the bias gradient is produced by the same launch (row sums of the dy tiles the
 * kernel stages anyway).  Query the workspace size (floats, may be 0) first.  Either of dw/db
 * may be NULL.                                                                      */
size_t pwg_conv1d_backward_weight_workspace_floats(const pwg_conv1d_desc* d);
int pwg_conv1d_backward_weight(const pwg_conv1d_desc* d, const float* x, const float* dy, float* dw,
                               float* db, float* workspace, size_t workspace_floats, void* stream);

/* Weight-normalised layers (old-style torch.nn.utils.weight_norm, dim 0: every conv of the generators and
 * discriminators): the same weight-gradient kernel, but the finishing kernel turns the reduction slabs
 * directly into dv and dg  (dg = <dw, v> / |v|,  dv = (g / |v|)(dw - v <dw, v> / |v|^2))  -- dw is never
 * materialised and the separate slab-reduction and weight-norm-backward launches disappear.  v / g: the
 * layer's weight_v (torch weight layout) and weight_g; db optional as above.  The workspace is mandatory. */
size_t pwg_conv1d_backward_weight_wn_workspace_floats(const pwg_conv1d_desc* d);
int pwg_conv1d_backward_weight_wn(const pwg_conv1d_desc* d, const float* x, const float* dy, const float* v,
                                  const float* g, float* dv, float* dg, float* db, float* workspace,
                                  size_t workspace_floats, void* stream);

/* Tuning / diagnostics: the same operation with an explicit tile configuration
 * (0 <= tile_config < pwg_conv1d_num_tile_configs()) and staging path (use_dma:
 * 1 = LDS-DMA double-buffered, 0 = register-staged).  tools/bench_conv.py sweeps
 * these to derive the heuristic inside pwg_conv1d_forward; results are identical
 * for every configuration up to fp32 summation order.                          */
int pwg_conv1d_num_tile_configs(void);
int pwg_conv1d_forward_cfg(const pwg_conv1d_desc* d, const float* x, const float* w_packed,
                           const float* bias, const float* add1, const float* add2, float* y,
                           int32_t tile_config, int32_t use_dma, void* stream);

/* Diagnostics (host only, no launch, no device needed): the plan pwg_conv1d_forward derives for a descriptor.
 * has_addends: 1 if add1 / add2 will be passed.  out[8]:
 *   out[0] kernel family: 0 = MFMA implicit-GEMM kernel, 1 = grouped 16x16x4 kernel, 2 = single-input-channel
 *          streaming kernel, 3 = few-output-channel streaming kernel (out[1..7] are 0 unless out[0] == 0)
 *   out[1] tile configuration, out[2] reduction slices (the workspace query is sized for them), out[3] 1 = LDS-DMA path
 *   out[4] logical tile order inside an XCD's run of workgroups: 0 = the row blocks of a column tile together (they
 *          share the x window in that XCD's L2; the default), 1 = the items of a (row block, reduction slice) together
 *          (they share the weight chunk; measured no faster, kept selectable: PWG_TILE_ORDER=1 forces it, =2 lets a
 *          bytes-per-XCD model choose per launch)
 *   out[5..7] grid (column tiles, row blocks x groups, items x slices).
 * pwg_debug_conv_tile_of_workgroup evaluates, on the host, the kernel's own dispatch-id -> logical-tile map
 * (out[3] = column tile, row block index, item * ksplit + slice): tests walk it to show it is a bijection.        */
int pwg_conv1d_plan(const pwg_conv1d_desc* d, int32_t has_addends, int32_t* out);
int pwg_debug_conv_tile_of_workgroup(int32_t grid_x, int32_t grid_y, int32_t grid_z, int32_t row_blocks,
                                     int32_t ksplit, int32_t item_major, int32_t workgroup, int32_t* out);

/* ------------------------------------------------------------------------- */
/* One HiFi-GAN MRF residual unit as ONE launch (inference; channels 32 / 64)  */
/*                                                                            */
/*   y = ( x + conv_{k,1}(lrelu(conv_{k,d}(lrelu(x)) + b1)) + b2 [+ add2] ) [/ out_div]      (has_conv2 = 1)
 *   y = ( x + conv_{k,d}(lrelu(x)) + b1 [+ add2] ) [/ out_div]                                (has_conv2 = 0)
 * replaces one iteration of the loop at layers/residual_block.py:253-257 (`xt = convs1[idx](x);
 * xt = convs2[idx](xt); x = xt + x`), add2 / out_div additionally fold the MRF mean of
 * models/hifigan.py:186-190.  All input channels of a column tile stay resident in LDS, the
 * intermediate activation never reaches HBM (csrc/resunit.hip).  Both convolutions are
 * "same"-padded with zeros (padding = (k-1)/2*dilation), k odd, t % 4 == 0, 0 < slope < 1;
 * pwg_resunit_supported() tells whether a unit fits (otherwise use two pwg_conv1d_forward calls:
 * identical result up to fp32 summation order).  Weights: torch layout (C, C, k) re-laid by
 * pwg_resunit_pack_weight (`scale` = optional weight-norm row scale as in pwg_conv1d_pack_weight).
 * x / y / add2: (batch, channels, t), 16-B aligned, y must not alias x.                            */
typedef struct pwg_resunit_desc {
  int32_t batch;
  int32_t channels;
  int32_t t;
  int32_t kernel;
  int32_t dilation;   /* of the first convolution; the second one has dilation 1 */
  int32_t has_conv2;
  float slope1;       /* LeakyReLU in front of the first convolution  */
  float slope2;       /* LeakyReLU in front of the second convolution */
  float out_div;      /* 1.0f = off */
} pwg_resunit_desc;
int pwg_resunit_supported(const pwg_resunit_desc* d);
/* 1 when the one-launch unit is also the faster choice on MI355X (measured, see csrc/resunit.hip) */
int pwg_resunit_profitable(const pwg_resunit_desc* d);
size_t pwg_resunit_packed_weight_floats(int32_t channels, int32_t kernel);
int pwg_resunit_pack_weight(int32_t channels, int32_t kernel, const float* w, const float* scale,
                            float* w_packed, void* stream);
int pwg_resunit_forward(const pwg_resunit_desc* d, const float* x, const float* w1_packed, const float* b1,
                        const float* w2_packed, const float* b2, const float* add2, float* y, void* stream);

/* ------------------------------------------------------------------------- */
/* One MelGAN residual stack as ONE launch (channels 48 / 96 / 192, kernel 3; csrc/resstack.hip)             */
/*                                                                            */
/*   h = conv_{3,dilation}(reflect_pad_dilation(lrelu(x))) + b1;   y = conv_{1x1}(lrelu(h)) + b2 + conv_{1x1}(x) + bs
 * replaces ResidualStack.forward (layers/residual_stack.py:75-85: `self.stack(c) + self.skip_layer(c)` with the
 * Sequential of :45-53 and the skip layer of :73).  All channels of a column tile stay resident in LDS; `h` (the
 * pre-activation output of the dilated convolution, what a backward pass needs) is written out only when the pointer
 * is given.  t % 4 == 0, t >= 64, 1 <= dilation <= 27, 0 < slope < 1; pwg_resstack_supported() tells whether a unit
 * fits (otherwise use three pwg_conv1d_forward calls: identical result up to fp32 summation order).  Weights: torch
 * layouts (C, C, 3), (C, C, 1), (C, C, 1) re-laid into ONE image by pwg_resstack_pack_weight (`scale*` = optional
 * weight-norm row scales as in pwg_conv1d_pack_weight).  x / y / h: (batch, channels, t), x 16-B aligned, y and h must
 * not alias x.                                                                                                   */
int pwg_resstack_supported(int32_t channels, int32_t t, int32_t dilation);
size_t pwg_resstack_packed_weight_floats(int32_t channels);
int pwg_resstack_pack_weight(int32_t channels, const float* w1, const float* scale1, const float* w2,
                             const float* scale2, const float* ws, const float* scale_s, float* w_packed, void* stream);
int pwg_resstack_forward(int32_t batch, int32_t channels, int32_t t, int32_t dilation, float slope, const float* x,
                         const float* w_packed, const float* b1, const float* b2, const float* bs, float* y, float* h,
                         void* stream);
/* Data gradient of the same unit in ONE launch (the backward of the three convolutions' data paths and of both
 * LeakyReLUs, layers/residual_stack.py:45-85 under autograd), in the padded domain of the dilated convolution:
 *   dh[t]  = lrelu'(h[t]) * (W2^T dy)[t]                                              (batch, channels, t)
 *   dxp[p] = lrelu'(xp[p]) * sum_tap (W1[tap]^T dh)[p - tap*dilation] + (Ws^T dy)[p - dilation]   (batch, channels, t + 2*dilation)
 * with xp = ReflectionPad1d(dilation)(x) and dy, dh zero outside [0, t): dx = pwg_pad1d_backward(dxp) (the
 * reflection's adjoint), dh is the dilated layer's weight-gradient operand.  `h` as written by pwg_resstack_forward;
 * weights re-laid (transposed) by pwg_resstack_pack_weight_bwd (same size as the forward image).                  */
int pwg_resstack_pack_weight_bwd(int32_t channels, const float* w1, const float* scale1, const float* w2,
                                 const float* scale2, const float* ws, const float* scale_s, float* w_packed,
                                 void* stream);
int pwg_resstack_backward_data(int32_t batch, int32_t channels, int32_t t, int32_t dilation, float slope,
                               const float* dy, const float* h, const float* x, const float* w_packed_bwd, float* dh,
                               float* dxp, void* stream);

/* One gated residual layer of the Parallel WaveGAN generator as ONE launch (csrc/wavenet.hip):
 *   z = conv_{k=3,dilation}(x) + b_dil + conv1x1_aux(c);  g = tanh(z[:64]) * sigmoid(z[64:]);
 *   skips_out = (conv1x1_skip(g) + b_skip + skips) * skip_mul;   x_out = (conv1x1_out(g) + b_out + x) * out_mul
 * replacing the five ATen calls of WaveNetResidualBlock.forward (layers/residual_block.py:102-140) plus the
 * running skip sum of the generator (models/parallel_wavegan.py:164-169).  Built for the PWG.v1 geometry
 * (64 residual / 128 gate / 64 skip / 80 aux channels, kernel 3, non-causal); pwg_wavenet_layer_supported says so.
 * x, skips, x_out, skips_out, g_out: (batch, 64, t); c: (batch, 80, t) (the upsampled mel); z_out: (batch, 128, t).
 * skips may be NULL (first layer); skips_out may alias skips; x_out must not alias x.  z_out / g_out (both or
 * neither; NULL in inference) receive the gate input and output for the backward pass.
 * `packed`: pwg_wavenet_pack_weights image of the four torch-layout weights w_dil (128, 64, 3), w_aux (128, 80, 1),
 * w_skip (64, 64, 1), w_out (64, 64, 1), each with an optional weight-norm row scale.                          */
typedef struct pwg_wavenet_desc {
  int32_t batch;
  int32_t t;
  int32_t residual_channels;
  int32_t gate_channels;
  int32_t skip_channels;
  int32_t aux_channels;
  int32_t kernel;
  int32_t dilation;
  int32_t causal;
  float out_mul;   /* sqrt(0.5) */
  float skip_mul;  /* 1, or sqrt(1 / layers) in the last layer */
} pwg_wavenet_desc;
int pwg_wavenet_layer_supported(const pwg_wavenet_desc* d);
size_t pwg_wavenet_packed_weight_floats(const pwg_wavenet_desc* d);
int pwg_wavenet_pack_weights(const pwg_wavenet_desc* d, const float* w_dil, const float* scale_dil, const float* w_aux,
                             const float* scale_aux, const float* w_skip, const float* scale_skip, const float* w_out,
                             const float* scale_out, float* packed, void* stream);
int pwg_wavenet_layer_forward(const pwg_wavenet_desc* d, const float* x, const float* c, const float* skips,
                              const float* packed, const float* b_dil, const float* b_skip, const float* b_out,
                              float* x_out, float* skips_out, float* z_out, float* g_out, void* stream);
/* Data path of the layer's backward pass in two launches (the weight gradients use pwg_conv1d_backward_weight*):
 *   gate_backward: dz (batch, 128, t) = d loss / d z from dx_out, ds_out (gradients w.r.t. x_out / skips_out; dx_out
 *     may be NULL) and the saved z -- the two 1x1 data gradients (K = 128), the out_mul / skip_mul scales and the
 *     tanh * sigmoid derivative; also writes go = out_mul * dx_out (the residual-path gradient and the out-conv
 *     weight-gradient operand; NULL with dx_out).
 *   data_backward: dx = dilated-conv data gradient of dz (+ go) and dc = aux 1x1 data gradient of dz (either may be NULL).
 * `packed_bwd`: pwg_wavenet_pack_weights_bwd image (depends on d->out_mul and d->skip_mul).                       */
size_t pwg_wavenet_packed_weight_bwd_floats(const pwg_wavenet_desc* d);
int pwg_wavenet_pack_weights_bwd(const pwg_wavenet_desc* d, const float* w_dil, const float* scale_dil, const float* w_aux,
                                 const float* scale_aux, const float* w_skip, const float* scale_skip, const float* w_out,
                                 const float* scale_out, float* packed, void* stream);
int pwg_wavenet_gate_backward(const pwg_wavenet_desc* d, const float* z, const float* dx_out, const float* ds_out,
                              const float* packed_bwd, float* dz, float* go, void* stream);
/* dc_accum (may be NULL, may alias dc): the gradient the LATER layers already accumulated for the shared aux features
 * c -- added in the epilogue, so the sum over the 30 layers needs no separate add launches.                          */
int pwg_wavenet_data_backward(const pwg_wavenet_desc* d, const float* dz, const float* go, const float* packed_bwd,
                              const float* dc_accum, float* dx, float* dc, void* stream);
/* The layer's parameter gradients in three launches: two contractions over time -- dz (128 rows) against the three
 * tap windows of x and c (272 rows), and [gs = skip_mul * ds_out ; go] (128 rows) against the saved gate output g (64
 * rows) -- into per-slice slabs, and one finish kernel that sums the slices in a fixed order and writes, per
 * convolution (grads[0..3] = dilated, aux, skip, out; torch weight layouts (rows, in, k)): the weight gradient (v NULL)
 * or, for a weight-normalised convolution w = g v / |v|, the gradients w.r.t. v and g; and the bias gradient.
 * go and grads[3].dw are NULL together (last layer); dg / db may be NULL.  Deterministic.                            */
typedef struct pwg_wavenet_param_grad {
  const float* v;  /* weight-norm direction tensor (shape of the weight), or NULL for a plain weight */
  const float* g;  /* weight-norm magnitudes (rows), NULL iff v is NULL */
  float* dw;       /* out: gradient w.r.t. the weight (v NULL) or w.r.t. v */
  float* dg;       /* out: gradient w.r.t. g, or NULL */
  float* db;       /* out: bias gradient, or NULL */
} pwg_wavenet_param_grad;
size_t pwg_wavenet_weight_backward_workspace_floats(const pwg_wavenet_desc* d);
int pwg_wavenet_weight_backward(const pwg_wavenet_desc* d, const float* dz, const float* x, const float* c, const float* gs,
                                const float* go, const float* g, const pwg_wavenet_param_grad* grads, float* workspace,
                                size_t workspace_floats, void* stream);

/* Old-style torch.nn.utils.weight_norm (dim=0) scale: scale[i] = g[i]/||v[i,...]||_2
 * replaces torch._weight_norm at every conv call site.
 * v: (n0, inner) flattened, g: (n0).                                          */
int pwg_weight_norm_scale(const float* v, const float* g, float* scale, int32_t n0,
                          int32_t inner, void* stream);
/* w = v * scale[dim0]  (materialises the torch-format weight, e.g. for
 * remove_weight_norm(): models/hifigan.py:209-219)                            */
int pwg_scale_rows(const float* v, const float* scale, float* w, int32_t n0, int32_t inner,
                   void* stream);

/* ------------------------------------------------------------------------- */
/* Activation / reparametrisation backward                                     */
/* ------------------------------------------------------------------------- */
/* dx = dy * scale * act'(.), act' written in terms of the activation OUTPUT y
 * (tanh: 1-y^2; leaky_relu: y>0 ? 1 : slope; relu: y>0).  Replaces the autograd
 * nodes of torch.nn.Tanh / LeakyReLU at models/hifigan.py:150, :329 etc.        */
int pwg_act_backward(const float* dy, const float* y, float* dx, int64_t n, int32_t act, float slope,
                     float scale, void* stream);
/* y = ((a + b) + c) / div (c may be NULL): the MRF combine `cs += block(c); c = cs / num_blocks`
 * (models/hifigan.py:186-190) when the residual blocks run as parallel graph branches.   */
int pwg_add3_div(const float* a, const float* b, const float* c, float* y, int64_t n, float div, void* stream);
/* Backward of old-style weight_norm (dim 0): given dw (torch layout) returns dv, dg. */
int pwg_weight_norm_backward(const float* dw, const float* v, const float* g, float* dv, float* dg,
                             int32_t n0, int32_t inner, void* stream);
/* torch.nn.utils.spectral_norm (dim 0, one power iteration, eps 1e-12) as used by the first
 * HiFi-GAN scale discriminator (models/hifigan.py:613-621,750-754).  w_orig viewed as
 * (rows, cols) = (c_out, c_in/groups * k).  do_iter != 0 updates u (rows) and v (cols) in
 * place (training-mode forward); sigma[0] = u^T W v; w = w_orig / sigma.
 * tmp: max(rows, 32 * cols) floats of workspace (row-sliced partial sums of W^T u). */
int pwg_spectral_norm_forward(const float* w_orig, float* u, float* v, float* sigma, float* w, float* tmp,
                              int32_t rows, int32_t cols, int32_t do_iter, float eps, void* stream);
/* The same, also leaving this forward's u / v in u_saved (rows) / v_saved (cols): the copies torch's hook takes for the
 * backward pass (torch/nn/utils/spectral_norm.py compute_weight: `u.clone()`, `v.clone()`), written by the iteration's own
 * kernels; fewer dependent launches (ABI v11).  u_saved == u / v_saved == v is allowed when do_iter == 0.  */
int pwg_spectral_norm_forward_saved(const float* w_orig, float* u, float* v, float* sigma, float* w, float* tmp,
                                    float* u_saved, float* v_saved, int32_t rows, int32_t cols, int32_t do_iter,
                                    float eps, void* stream);
/* dw_orig = dw / sigma - (<dw, w_orig> / sigma^2) u v^T ;  scratch: PWG_SPECTRAL_NORM_SCRATCH_FLOATS floats
 * (the dot product is summed through per-block shares in a fixed order: no atomics).  */
#define PWG_SPECTRAL_NORM_SCRATCH_FLOATS 257
int pwg_spectral_norm_backward(const float* dw, const float* w_orig, const float* u, const float* v,
                               const float* sigma, float* dw_orig, float* scratch, int32_t rows,
                               int32_t cols, void* stream);

/* ------------------------------------------------------------------------- */
/* Parallel WaveGAN specific element-wise stages                               */
/* ------------------------------------------------------------------------- */
/* WaveNet gate: out = tanh(z[:, :C]) * sigmoid(z[:, C:]); z (B, 2C, T), out (B, C, T)
 * (layers/residual_block.py:120-132).                                           */
int pwg_gate_forward(const float* z, float* out, int32_t batch, int32_t channels, int64_t t, void* stream);
int pwg_gate_backward(const float* z, const float* dout, float* dz, int32_t batch, int32_t channels,
                      int64_t t, void* stream);
/* One stage of the mel upsampler: F.interpolate(nearest, x scale) followed by the
 * (freq_kernel, 2*scale+1) single-channel Conv2d over (mel channel, time), fused
 * (layers/upsample.py:43-45,88-103,121-127):
 *   y[b][c][t] = sum_f sum_j w[f][j] * x[b][c + f - (F-1)/2][(t + j - pad_left) / scale],
 * rows = B * channels (zero padding on the channel axis; freq_kernel = 1 is the shipped recipes' case);
 * pad_left = scale (centred) or 2*scale (use_causal_conv: :96-99,121-125, output trimmed
 * to the stretched length).  `act` / `slope`: the optional nonlinearity after the stage (:105-110), applied in the
 * epilogue; its backward is pwg_act_backward on the stage output, then pwg_stretch_conv_backward.  The weight
 * gradient is summed through per-block shares in `workspace` in a fixed order (no atomics).           */
#define PWG_STRETCH_CONV_WGRAD_BLOCKS 512
int pwg_stretch_conv_forward(const float* x, const float* w, float* y, int64_t rows, int32_t t_in,
                             int32_t scale, int32_t kernel, int32_t pad_left, int32_t channels,
                             int32_t freq_kernel, int32_t act, float slope, void* stream);
size_t pwg_stretch_conv_backward_workspace_floats(int32_t kernel, int32_t freq_kernel);
int pwg_stretch_conv_backward(const float* dy, const float* x, const float* w, float* dx, float* dw,
                              int64_t rows, int32_t t_in, int32_t scale, int32_t kernel, int32_t pad_left,
                              int32_t channels, int32_t freq_kernel, float* workspace, size_t workspace_floats,
                              void* stream);
/* Pseudo-QMF filterbank as polyphase kernels (layers/pqmf.py:120-149; csrc/pqmf.hip), 1 <= subbands <= 8:
 *   down:  y[b][k][i] = sum_j  h[k][j]               * x[b][i K + j - pad]     x (B, 1, t_in) -> y (B, K, n_out)
 *   up:    x[b][t]    = sum_k sum_i g[k][t + pad - i K] * y[b][k][i]           y (B, K, n_in) -> x (B, 1, t_out)
 * (zero outside the signals).  PQMF.analysis = down with h = analysis_filter, pad = taps / 2,
 * n_out = floor(T / K) for ANY T (the reference's stride-K pick after the full-rate FIR, pqmf.py:120-131);
 * PQMF.synthesis = up with g[k][m] = K * synthesis_filter[k][taps - m], t_out = K * n_in (pqmf.py:133-149).
 * Each is the adjoint of the other with the same filter, which is how the backward passes run.  */
int pwg_pqmf_down(const float* x, const float* h, float* y, int32_t batch, int64_t t_in, int64_t n_out,
                  int32_t subbands, int32_t len, int32_t pad, void* stream);
int pwg_pqmf_up(const float* y, const float* g, float* x, int32_t batch, int64_t n_in, int64_t t_out,
                int32_t subbands, int32_t len, int32_t pad, void* stream);

/* ---- StyleMelGAN pieces (layers/tade_res_block.py, models/style_melgan.py) ---- */
/* torch.nn.InstanceNorm1d(affine=False) over `rows` = B*C rows of t samples (tade_res_block.py:27,66):
 * y = (x - mean) / sqrt(var + eps) with the biased variance; mean / rstd (rows floats each) are kept
 * for the backward: dx = rstd * (dy - mean(dy) - y * mean(dy * y)).                              */
int pwg_instance_norm_forward(const float* x, float* y, float* mean, float* rstd, int64_t rows, int32_t t,
                              float eps, void* stream);
int pwg_instance_norm_backward(const float* dy, const float* y, const float* rstd, float* dx, int64_t rows,
                               int32_t t, void* stream);
/* torch.nn.Upsample(scale_factor, mode="nearest") along time (+ optional addend of the output shape,
 * the `upsample(residual) + x` of tade_res_block.py:161): y[r][t] = x[r][t / scale] (+ add[r][t]). */
int pwg_upsample_nearest_forward(const float* x, const float* add, float* y, int64_t rows, int32_t t_in,
                                 int32_t scale, void* stream);
int pwg_upsample_nearest_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t scale,
                                  void* stream);
/* TADE modulation (tade_res_block.py:69-72): cg (B, 2C, t_in*scale), xn (B, C, t_in):
 * y[b][c][t] = cg[b][c][t] * xn[b][c][t / scale] + cg[b][C + c][t].                              */
int pwg_tade_modulate_forward(const float* xn, const float* cg, float* y, int32_t batch, int32_t channels,
                              int32_t t_in, int32_t scale, void* stream);
int pwg_tade_modulate_backward(const float* dy, const float* xn, const float* cg, float* dxn, float* dcg,
                               int32_t batch, int32_t channels, int32_t t_in, int32_t scale, void* stream);
/* Gated activation of TADEResBlock (tade_res_block.py:151-158): z (B, 2C, T) -> y (B, C, T),
 * y = softmax_over_channels(z[:, :C]) * tanh(z[:, C:])  (use_softmax != 0) or sigmoid(.) * tanh(.). */
int pwg_softmax_gate_forward(const float* z, float* y, int32_t batch, int32_t channels, int64_t t,
                             int32_t use_softmax, void* stream);
int pwg_softmax_gate_backward(const float* z, const float* dy, float* dz, int32_t batch, int32_t channels,
                              int64_t t, int32_t use_softmax, void* stream);

/* ---- UHiFiGAN pieces (models/uhifigan.py) ---- */
/* One operand of torch.cat(dim=1) (uhifigan.py:286): y (batch, c_dst, t)[:, c_off : c_off + c_src] = x
 * (batch, c_src, t); reverse != 0 copies that slice of y back into x (the concatenation's backward). */
int pwg_copy_channels(float* x, float* y, int32_t batch, int32_t c_src, int32_t c_dst, int32_t c_off,
                      int64_t t, int32_t reverse, void* stream);
/* torch.nn.Dropout in training mode (uhifigan.py:86,130): y = x * keep / (1 - p) with a counter-based
 * mask hash(seed + *seed_dev, index) >= p * 2^32; calling it again with the same seeds on the output
 * gradient is the backward.  seed_dev (optional device scalar) lets a captured hipGraph draw a new mask
 * at every replay.  (Different random stream than torch's Philox: same distribution, other draws.)    */
int pwg_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev,
                void* stream);

/* ---- training input assembly (bin/train.py:646-896 Collater, mel -> waveform branch) ---- */
/* Random-crop batch from a device-resident corpus: audio / mel are the utterances concatenated
 * (mel row-major (frames, channels)), *_off their start offsets (elements / frames), audio_len the
 * waveform lengths.  For item b of utterance utt[b] and start frame start[b]:
 *   y (batch, 1, steps)                 = audio[start*hop : start*hop + steps]   (edge-padded)
 *   c (batch, channels, frames_ctx)     = mel[start - acw : start - acw + frames_ctx]^T
 * with frames_ctx = steps/hop + 2*acw.  The start frames are drawn by the caller.     */
int pwg_gather_crop(const float* audio, const int64_t* audio_off, const int64_t* audio_len, const float* mel,
                    const int64_t* mel_off, const int32_t* utt, const int32_t* start, float* y, float* c,
                    int32_t batch, int32_t steps, int32_t hop, int32_t frames_ctx, int32_t acw,
                    int32_t channels, void* stream);

/* ---- inference I/O helpers (around bin/decode.py:214-243) ---- */
/* pcm[i] = (int16) rint(clamp(x[i], -1, 1) * 32767): the PCM_16 conversion of the synthesised wave. */
int pwg_wave_to_pcm16(const float* x, int16_t* pcm, int64_t n, void* stream);
/* y (batch, channels, frames) = (x (batch, frames, channels) - mean[c]) / scale[c]: the
 * `normalize_before` step of `inference` (models/hifigan.py:262-266) fused with its transpose;
 * mean/scale may both be NULL (transpose only).                                       */
int pwg_normalize_transpose(const float* x, const float* mean, const float* scale, float* y,
                            int32_t batch, int32_t frames, int32_t channels, void* stream);

/* ------------------------------------------------------------------------- */
/* Pooling / explicit padding                                                  */
/* ------------------------------------------------------------------------- */
/* torch.nn.AvgPool1d over `rows` rows (models/hifigan.py:773-775: k4 s2 p2,
 * count_include_pad=1; models/melgan.py:409-414: k4 s2 p1 count_include_pad=0).   */
int pwg_avg_pool1d_forward(const float* x, float* y, int64_t rows, int32_t t_in, int32_t t_out,
                           int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad,
                           void* stream);
int pwg_avg_pool1d_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t t_out,
                            int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad,
                            void* stream);
/* F.pad(x, (pad_left, pad_right), mode) for mode = PWG_PAD_* (period discriminator tail,
 * models/hifigan.py:365-368; MelGAN ReflectionPad1d when training).                */
int pwg_pad1d_forward(const float* x, float* y, int64_t rows, int32_t t_in, int32_t pad_left,
                      int32_t pad_right, int32_t mode, void* stream);
int pwg_pad1d_backward(const float* dy, float* dx, int64_t rows, int32_t t_in, int32_t pad_left,
                       int32_t pad_right, int32_t mode, void* stream);

/* ------------------------------------------------------------------------- */
/* Spectral losses: torch.stft(center=True, reflect) is computed as            */
/*   frame_fold  (B,T) -> (B, hop, n_cols):  y[b][c][n] = reflect_pad(x)[n*hop+c] */
/*   conv1d      Cin = hop, Cout = 2*bins, k = ceil(win/hop)  (windowed DFT basis  */
/*               as the weight; runs on the MFMA conv kernel above)               */
/*   stft_mag    sqrt(max(re^2+im^2, eps))                                        */
/* replacing losses/stft_loss.py:16-40 and losses/mel_loss.py:95-110.             */
/* ------------------------------------------------------------------------- */
int pwg_frame_fold_forward(const float* x, float* y, int32_t batch, int32_t t, int32_t pad, int32_t hop,
                           int32_t n_cols, void* stream);
int pwg_frame_fold_backward(const float* dy, float* dx, int32_t batch, int32_t t, int32_t pad,
                            int32_t hop, int32_t n_cols, void* stream);
/* spec: (B, 2*bins, frames), rows [0,bins) real, [bins,2*bins) imaginary.         */
int pwg_stft_mag_forward(const float* spec, float* mag, int32_t batch, int32_t bins, int32_t frames,
                         float eps, void* stream);
int pwg_stft_mag_backward(const float* spec, const float* mag, const float* dmag, float* dspec,
                          int32_t batch, int32_t bins, int32_t frames, float eps, void* stream);
/* y = log(max(x, eps)) / log_div  (log_div = 1, ln2, ln10: losses/mel_loss.py:71-79,108-110) */
int pwg_log_clamp_forward(const float* x, float* y, int64_t n, float eps, float log_div, void* stream);
int pwg_log_clamp_backward(const float* x, const float* dy, float* dx, int64_t n, float eps, float log_div,
                           void* stream);

/* Fused single-resolution STFT loss of a (predicted, target) pair -- one launch instead of the
 * frame/conv/mag/log/reduce chain above, and no spectrum-shaped tensor in HBM:
 *   sums[0] = sum (|Y| - |X|)^2   sums[1] = sum |Y|^2   sums[2] = sum |log|Y| - log|X||
 * over the (B, bins, frames) magnitudes |.| = sqrt(max(re^2 + im^2, eps)), and the two losses themselves:
 *   sums[3] = SpectralConvergenceLoss = sqrt(sums[0]) / sqrt(sums[1])      (losses/stft_loss.py:61)
 *   sums[4] = LogSTFTMagnitudeLoss    = sums[2] / (B * bins * frames)      (:82)
 * (`sums`: 5 device floats); torch.stft + clamp + sqrt are :16-40.
 *   fx, fy : the two signals folded by pwg_frame_fold_forward, (B, hop, n_cols), n_cols >= frames + taps - 1
 *   basis  : windowed DFT image [tap][hop][m_pad], m_pad = 64 * ceil(bins / 32); each 64-row group holds
 *            32 cosine rows then the 32 matching -sine rows (rows of bins >= `bins` are zero)
 *   workspace: pwg_stft_loss_workspace_floats() floats.  Deterministic (fixed tiles, fixed sum order).
 * Backward: given the forward's `sums` and g2 = d loss / d (sums[3], sums[4]) (2 DEVICE floats) writes dspec
 * (B, 2*bins, frames) = [d re | d im]  (the gradient of sums[3] is 0 where sums[0] == 0, torch.norm's subgradient)
 * of the PREDICTED signal's spectrum (the target is a constant), i.e. the dy operand of
 * pwg_conv1d_backward_data on the DFT convolution; pwg_frame_fold_backward finishes the chain.        */
size_t pwg_stft_loss_workspace_floats(int32_t batch, int32_t bins, int32_t frames);
int pwg_stft_loss_forward(const float* fx, const float* fy, const float* basis, int32_t batch, int32_t hop,
                          int32_t n_cols, int32_t taps, int32_t bins, int32_t frames, float eps,
                          float* workspace, float* sums, void* stream);
int pwg_stft_loss_backward(const float* fx, const float* fy, const float* basis, int32_t batch, int32_t hop,
                           int32_t n_cols, int32_t taps, int32_t bins, int32_t frames, float eps,
                           const float* sums, const float* g2, float* dspec, void* stream);

/* ---- STFT loss through a radix-2 FFT in LDS (round 4; power-of-two n_fft in 256 .. 2048) ----
 * Same contract as pwg_stft_loss_forward / _backward (reference losses/stft_loss.py:16-40, :61, :82: torch.stft with
 * center=True / reflect padding / window zero-padded to n_fft, clamp 1e-7, sqrt, the two losses), computed from the raw
 * signals x (predicted), y (target), both (B, T): one complex FFT per frame carries both signals.  window: `win` floats;
 * twiddle: n_fft / 2 pairs (cos, -sin)(2 pi t / n_fft) (host float64 -> float32); sums: 5 floats [S_d, S_y, S_l, sc, mag].
 * Backward: g2 = upstream gradients of (sc, mag) on the device; dframes: scratch of B * (1 + T / hop) * win floats; dx (B, T). */
int pwg_stft_fft_supported(int32_t n_fft, int32_t win, int32_t hop);
size_t pwg_stft_fft_workspace_floats(int32_t batch, int32_t frames, int32_t n_fft);
int pwg_stft_fft_loss_forward(const float* x, const float* y, const float* window, const float* twiddle, int32_t batch,
                              int32_t t, int32_t n_fft, int32_t hop, int32_t win, float eps, float* workspace,
                              float* sums, void* stream);
int pwg_stft_fft_loss_backward(const float* x, const float* y, const float* window, const float* twiddle, int32_t batch,
                               int32_t t, int32_t n_fft, int32_t hop, int32_t win, float eps, const float* sums,
                               const float* g2, float* dframes, float* dx, void* stream);

/* Mel-spectrogram loss through the same FFT (reference losses/mel_loss.py:95-110, :150-165; eps = the module's 1e-10):
 * total[0] = sum_{b,j,f} | log(max(mel_x, eps)) - log(max(mel_y, eps)) | / log_div.  fb: filterbank [n_mels][bins_pad]
 * (bin fastest, bins_pad >= n_fft / 2 + 1); mel_range: per mel (first, last) bin of its non-zero support; bin_range: per
 * bin (first, last) mel covering it (int32 pairs).  workspace: pwg_stft_fft_workspace_floats() floats; dframes / dx as above. */
int pwg_mel_fft_loss_forward(const float* x, const float* y, const float* window, const float* twiddle, const float* fb,
                             const int32_t* mel_range, const int32_t* bin_range, int32_t batch, int32_t t, int32_t n_fft,
                             int32_t hop, int32_t win, int32_t n_mels, int32_t bins_pad, float eps, float log_div,
                             float* workspace, float* total, void* stream);
int pwg_mel_fft_loss_backward(const float* x, const float* y, const float* window, const float* twiddle, const float* fb,
                              const int32_t* mel_range, const int32_t* bin_range, int32_t batch, int32_t t, int32_t n_fft,
                              int32_t hop, int32_t win, int32_t n_mels, int32_t bins_pad, float eps, float log_div,
                              const float* gout, float* dframes, float* dx, void* stream);

/* Fused mel-spectrogram loss of a (predicted, target) pair (losses/mel_loss.py:95-110,150-165):
 *   sum[0] = sum_{b,j,f} | log(max(mel_x, eps)) - log(max(mel_y, eps)) | / log_div,
 *   mel = filterbank (n_mels x bins) applied to |STFT| = sqrt(max(re^2 + im^2, eps)); F.l1_loss is sum[0] / (B * n_mels * frames).
 * Same tiling and operands as pwg_stft_loss_*; the filterbank contraction runs on MFMA from the magnitude
 * registers.  mel_t: filterbank [32 * ceil(bins / 32)][mels_pad] (mel fastest, mels_pad = 32 * ceil(n_mels / 32),
 * zero padded); mel_b: the same matrix as [mels_pad][32 * ceil(bins / 32)] (bin fastest).  mel_x / mel_y
 * (B, n_mels, frames) receive the un-clamped mels (kept for the backward pass).  workspace:
 * pwg_mel_loss_workspace_floats() floats.  Backward writes dspec = [d re | d im] of the predicted signal.   */
size_t pwg_mel_loss_workspace_floats(int32_t batch, int32_t bins, int32_t frames, int32_t n_mels);
int pwg_mel_loss_forward(const float* fx, const float* fy, const float* basis, const float* mel_t, int32_t batch,
                         int32_t hop, int32_t n_cols, int32_t taps, int32_t bins, int32_t frames, int32_t n_mels,
                         float eps, float log_div, float* workspace, float* mel_x, float* mel_y, float* sum,
                         void* stream);
int pwg_mel_loss_backward(const float* fx, const float* basis, const float* mel_b, const float* mel_x,
                          const float* mel_y, int32_t batch, int32_t hop, int32_t n_cols, int32_t taps,
                          int32_t bins, int32_t frames, int32_t n_mels, float eps, float log_div,
                          const float* gout, float* dspec, void* stream);

/* ------------------------------------------------------------------------- */
/* Loss reductions (deterministic two-stage sums)                              */
/*   out[0] = scale * sum_i term_i ;  mode 0 |a-b| (F.l1_loss: feat_match_loss.py:44, */
/*   mel_loss.py:163, stft_loss.py:82), 1 (a-b)^2 and 2 a^2 (Frobenius norms,        */
/*   stft_loss.py:61), 3 (a-c)^2 (F.mse_loss against a constant: adversarial_loss.py */
/*   :54-58,113-123), 4 a.  workspace: >= 512 floats.                                */
/* ------------------------------------------------------------------------- */
/*   5 -min(a-1, 0) and 6 -min(-a-1, 0): the hinge terms of DiscriminatorAdversarialLoss            */
/*   (adversarial_loss.py:119-123; ties of the min get gradient 1/2 like torch.minimum).            */
enum { PWG_RED_ABS_DIFF = 0, PWG_RED_SQ_DIFF = 1, PWG_RED_SQ = 2, PWG_RED_SQ_DIFF_CONST = 3, PWG_RED_SUM = 4,
       PWG_RED_HINGE_REAL = 5, PWG_RED_HINGE_FAKE = 6 };
int pwg_reduce_forward(const float* a, const float* b, float c, int64_t n, int32_t mode, float scale,
                       float* out, float* workspace, void* stream);
/* da = gout[0] * scale * dterm/da (db = -da); gout is a DEVICE scalar.            */
int pwg_reduce_backward(const float* a, const float* b, float c, int64_t n, int32_t mode, float scale,
                        const float* gout, float* da, float* db, void* stream);

/* Multi-tensor form: the Python loops over discriminators x layers of FeatureMatchLoss.forward
 * (feat_match_loss.py:36-54) and of the adversarial losses (adversarial_loss.py:30-47,79-104) as ONE
 * launch pair:  out[slot] (+)= sum_items scale_item * sum_i term_mode(a_i, b_i | c).
 * `items` is a HOST array (<= PWG_RED_MAX_ITEMS entries; it is copied into the kernel arguments, so
 * a captured hipGraph keeps its own copy); workspace: pwg_multi_reduce_workspace_floats() floats.
 * accumulate != 0 adds to out[] (more than PWG_RED_MAX_ITEMS tensors = several calls).
 * Backward: da = gout[slot] * scale * dterm/da and db = -da into the items' da / db (NULL = skip). */
#define PWG_RED_MAX_ITEMS 64
typedef struct pwg_red_item {
  const float* a;
  const float* b;  /* second operand of the diff modes, else NULL                    */
  float* da;       /* backward only: gradient w.r.t. a (NULL = not needed)           */
  float* db;       /* backward only: gradient w.r.t. b (NULL = not needed)           */
  int64_t n;       /* elements                                                       */
  int32_t mode;    /* PWG_RED_*                                                      */
  int32_t slot;    /* which output scalar this item adds to                          */
  float scale;     /* e.g. 1/n for a mean, times the averaging weights               */
  float c;         /* constant of PWG_RED_SQ_DIFF_CONST                              */
} pwg_red_item;
size_t pwg_multi_reduce_workspace_floats(const pwg_red_item* items, int32_t n_items);
int pwg_multi_reduce_forward(const pwg_red_item* items, int32_t n_items, int32_t n_slots, float* out,
                             int32_t accumulate, float* workspace, void* stream);
int pwg_multi_reduce_backward(const pwg_red_item* items, int32_t n_items, int32_t n_slots, const float* gout,
                              void* stream);

/* ------------------------------------------------------------------------- */
/* Fused multi-tensor optimizer steps over a device table of chunks            */
/* (replaces the per-parameter Python loops of torch.optim.Adam and            */
/* optimizers/radam.py:27-99)                                    */
/* ------------------------------------------------------------------------- */
typedef struct pwg_opt_chunk {
  float* p;        /* parameter slice                                    */
  const float* g;  /* gradient slice                                     */
  float* m;        /* exp_avg                                            */
  float* v;        /* exp_avg_sq                                         */
  float* vmax;     /* max_exp_avg_sq (amsgrad) or NULL                   */
  int32_t n;       /* elements in this slice                             */
  int32_t pad_;
} pwg_opt_chunk;
/* `step` is the 1-based step count AFTER increment, as torch uses it for bias correction;
 * gradients are multiplied by grad_scale first (1/world_size after a sum all-reduce).
 * pwg_adam_step*: weight_decay > 0 is torch.optim.Adam's L2 form (g += wd * p); weight_decay < 0 selects
 * torch.optim.AdamW's DECOUPLED decay of magnitude |weight_decay| (p *= 1 - lr * |wd| before the update).   */
int pwg_adam_step(const void* chunks, int32_t n_chunks, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, void* stream);
int pwg_radam_step(const void* chunks, int32_t n_chunks, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int32_t step, float grad_scale, void* stream);
/* hipGraph-friendly variants: the scalars come from 8 floats in DEVICE memory
 *   [lr, beta1, beta2, eps, weight_decay, step_size, aux, grad_scale]
 * (Adam: step_size = lr/(1-beta1^t), aux = sqrt(1-beta2^t); RAdam: step_size per radam.py:63-86
 * incl. lr, aux = 1 when rectified), refreshed by the host between graph replays.     */
int pwg_adam_step_dev(const void* chunks, int32_t n_chunks, const float* hyper, void* stream);
int pwg_radam_step_dev(const void* chunks, int32_t n_chunks, const float* hyper, void* stream);
/* torch.nn.utils.clip_grad_norm_ (bin/train.py:289-293,329-333): out[0] = total L2 norm,
 * out[1] = applied coefficient; gradients are scaled in place.  workspace >= n_chunks floats. */
int pwg_clip_grad_norm(const void* chunks, int32_t n_chunks, float max_norm, float* out, float* workspace,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PWG_KERNELS_H_ */
