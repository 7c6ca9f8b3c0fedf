/*
 * dvslam.h -- C-ABI of libdvslam_hip.so: the MI355X (gfx950) hot path of the Monodepth2-style VO
 * training step of chansoopark98/Deep-Visual-SLAM.
 *
 * The reference has no FFI/plugin layer: its boundary is the Python operator surface
 * (model/layers.py, vo/learner_func.py, vo/learner_new.py, model/{resnet_encoder,depthnet,
 * posenet_single}.py).  This library sits directly under that surface; each entry point cites the
 * reference code (file:line under the reference repo) whose arithmetic it replaces.  The binding a
 * maintainer adds on the reference side is a ctypes stub, shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative dvs_status on error; dvs_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - all pointers are DEVICE pointers to fp32, NCHW-contiguous buffers owned by the caller unless
 *     a parameter is documented as host; nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous,
 *     re-entrant and keep no thread-local state besides the error string (the reference's
 *     backward runs on PyTorch's autograd thread);
 *   - sizes are ints; B = batch, H/W = full-resolution image size.
 */
#ifndef DVSLAM_H
#define DVSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DVS_OK = 0,
    DVS_ERR_INVALID = -1,   /* bad argument (null pointer, non-positive size, unsupported shape) */
    DVS_ERR_LAUNCH = -2,    /* hipLaunch / runtime failure, see dvs_last_error() */
    DVS_ERR_UNSUPPORTED = -3
} dvs_status;

#define DVS_MAX_SCALES 4

/* ---------------------------------------------------------------------------------------------
 * library
 * ------------------------------------------------------------------------------------------- */
const char* dvs_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int dvs_abi_version(void);
/* Name of the code-object architecture the kernels were compiled for ("gfx950"). */
const char* dvs_arch(void);

/* Deterministic forward (test / debugging mode, off by default): with on != 0 no forward or data-gradient launch is split
 * over K with float atomics and the BatchNorm partial sums are added in one fixed order, so a forward pass repeats bit
 * for bit and two runs take the same ReLU / maxpool branches (their gradients then differ by smooth rounding noise only;
 * DESIGN.md section 6).  The BatchNorm batch statistics must then come from dvs_bn_bwd_reduce(dz = y, z = NULL, y,
 * mean = 0, invstd = 1) instead of the convolution's atomic statistics epilogue (the Python side does that). */
int dvs_set_deterministic(int on);
int dvs_get_deterministic(void);

/* Arithmetic of the implicit-GEMM convolutions (ABI 8; process-wide, 0 by default).  mode 0: fp32 operands on the fp32 matrix
 * cores -- the reference's default precision and what every parity statement of this library is made for.  mode 1: bf16 operands
 * (rounded to nearest-even as they are staged into LDS), fp32 accumulation, fp32 tensors in HBM on both sides -- the counterpart
 * of the reference's `use_amp` branch (vo/train.py:44,177-185: torch.autocast + GradScaler around the same modules); forward,
 * data and weight gradient of dvs_conv2d_* take it (the stems on bf16 kernels of their own), the dvs_conv3x3_bf16_* entry points
 * below exist for it; the Winograd kernels (not used in the mode), the row-ring kernels of the 16-channel layers, the heads,
 * BatchNorm, the loss chain and the optimiser stay fp32.  Results then differ from mode 0 by bf16 operand rounding (2^-9 relative per product, averaging down over
 * K): separately toleranced in tests/test_bf16_gpu.py, never the headline precision. */
int dvs_set_precision(int mode);
int dvs_get_precision(void);

/* Peak probes (ABI 7; measurement aids of bench.py's `measured_peaks`, timed by the caller with events on `stream`):
 *   dvs_peak_probe_mfma: one wave per SIMD on every CU runs 4 * iters independent v_mfma_f32_32x32x2_f32 from registers; *flops
 *                        (host) = the flops the launch executes.  scratch: any device buffer of >= 4 bytes (never written).
 *   dvs_peak_probe_copy: a 16-byte-per-lane streaming copy of `bytes` (multiple of 16) from src to dst: 2 * bytes of HBM traffic. */
int dvs_peak_probe_mfma(float* scratch, int iters, double* flops, void* stream);
int dvs_peak_probe_copy(const void* src, void* dst, size_t bytes, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream around each main kernel (used by
 * bench.py for the roofline line; off by default).  dvs_profile_enable(1) clears the counters;
 * dvs_profile_read synchronises the recorded events of one slot and returns the accumulated
 * kernel time and launch count.  Slot names are the kernel names seen by rocprofv3. */
int dvs_profile_enable(int on);
int dvs_profile_slots(void);
const char* dvs_profile_slot_name(int slot);
int dvs_profile_read(int slot, double* total_ms, long* launches);
/* Algorithmic work of the launches recorded in a slot: flops for the MFMA kernels (2*M*N*K of each
 * convolution), 0 for kernels whose byte count the caller derives from the problem size. */
int dvs_profile_work(int slot, double* work);

/* ---------------------------------------------------------------------------------------------
 * a13  Adam over a flat fp32 arena: one pass replaces torch.optim.Adam(params, lr) + zero_grad of
 *      the reference trainer (vo/train.py:114-117,175,192).  torch.optim.Adam arithmetic (no weight
 *      decay, no amsgrad): m,v updates, bias corrections, p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2)+eps).
 *      grad is multiplied by grad_scale first (1/world_size after a sum all-reduce); zero_grad != 0
 *      clears grad in the same pass.  All four buffers hold n floats and are 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int dvs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1-a3  convolutions of ResnetEncoder / DepthNet decoder / PoseNet decoder as implicit GEMMs on the
 *        fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 *        replaces nn.Conv2d inside torchvision BasicBlock (model/resnet_encoder.py:83-111),
 *        Conv3x3 / ConvBlock (model/layers.py:106-136), the decoder's upsample + concat
 *        (model/depthnet.py:79-88, model/layers.py:196-199) and PoseNet's head
 *        (model/posenet_single.py:174-202).
 *   Layout: activations NHWC (torch channels_last of the reference's NCHW tensors), weights
 *   [Cout][kh][kw][Cin] (torch channels_last of the reference's [Cout,Cin,kh,kw] parameter).
 *   Output size: Ho = (H + 2 pad - kh)/stride + 1 (same for W).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int B, H, W, Cin, Cout;
    int kh, kw, stride;
    int pad;        /* implicit padding on every side */
    int pad_mode;   /* 0 = zeros (nn.Conv2d padding), 1 = nn.ReflectionPad2d(pad) in front of the conv;
                       dvs_conv2d_dgrad only: 2 = H x W is an already reflection-padded input, pad = 0 (dvs_reflect_fold follows) */
} dvs_conv_desc;

typedef struct {
    /* input-side fusion (applied while the im2col slice is gathered) */
    const float* x2;        /* non-NULL: input = concat(upsample_nearest2x(x [B,H/2,W/2,C1]), x2 [B,H,W,Cin-C1]) */
    int C1;
    const float* in_scale;  /* non-NULL: x <- x * in_scale[c] + in_shift[c] (folded BatchNorm / input normalisation) */
    const float* in_shift;
    int in_relu;            /* then ReLU */
    int nchw_planar;        /* x is a planar [B,Cin,H,W] image (encoder conv1); any Cin */
    /* output-side fusion */
    int act;                /* 0 none, 1 ReLU, 2 ELU(alpha=1), 3 sigmoid, 4 GELU (erf form; forward only: the ViT MLP of
                               model/depth_anything_v2/dinov2_layers/mlp.py:35-41) */
    float* stats;           /* non-NULL: [2][Cout], += per-channel sum and sum of squares of the raw
                               (pre-bias, pre-activation) output: BatchNorm batch statistics */
    int stat_groups;        /* 0 / 1: one set of statistics; 2: stats is [2][2][Cout], the first and second half of
                               the batch counted separately -- PoseNet runs its two frame pairs as one batch of 2B
                               and the reference's two calls (vo/learner_new.py:113-114) normalise each pair alone */
    const float* residual;  /* non-NULL: [B,Ho,Wo,Cout] added before the activation: y = act(conv + bias + residual) --
                               the BasicBlock tail relu(bn2(conv2(.)) + identity) with an eval-mode BatchNorm folded
                               into w / bias (inference path of model/resnet_encoder.py:100-111) */
    int stat_slots;         /* (ABI 5) 0 / 1: one table; a power of two <= 64: stats is [stat_slots][G][2][Cout] and output tile t
                               adds into copy t % stat_slots -- a 1x1 downsample convolution is all epilogue, and ~900 tiles
                               adding to the same 2 x Cout addresses serialise in the L2; dvs_bn_fwd_slots adds the copies up.
                               Not for the planar (conv1) input. */
} dvs_conv_fusion;

/* y [B,Ho,Wo,Cout] = act(conv(x, w) + bias); `f` may be NULL (no fusion). */
int dvs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const dvs_conv_desc* d,
                   const dvs_conv_fusion* f, void* stream);

/* Backward of dvs_conv2d_fwd.
 *   dvs_conv2d_pack_wt: wt[ci][kh][kw][co] = w[co][kh][kw][ci], the [N][K] operand of the data gradient.
 *   dvs_conv2d_dgrad:   dx [B,H,W,Cin] = conv_transpose(dy * act'(y_out), w).  `d` describes the FORWARD
 *                       conv; stride 1 or 2; pad_mode 1 only for ReflectionPad2d(1) + 3x3 (the reflection's
 *                       gradient fold is done in the gather).  y_out/dact: forward output and its
 *                       activation code (0 = dy is already the pre-activation gradient).  For an
 *                       upsample+concat forward, dx is the gradient of the concatenated [B,H,W,Cin] input.
 *   dvs_conv2d_wgrad:   dw [Cout][Ktot] += sum_pixels (dy * act'(y_out)) x im2col(x), dbias [Cout] += column
 *                       sums (NULL = skip); both are accumulated with atomics, the caller zero-fills.  `f`
 *                       carries the forward's input-side fusion (x2/C1, in_scale/in_shift/in_relu,
 *                       nchw_planar; act/stats ignored).  With nchw_planar dw is packed [Cout][Cin][kh][8].
 *   Channel counts must be multiples of 4 (NHWC 16-byte gathers). */
/* Winograd F(2x2, 3x3) for the stride-1 / pad-1 3x3 convolutions of the ResNet BasicBlocks (torchvision BasicBlock conv1/conv2
 * as used by model/resnet_encoder.py:94-111): 16 products per 2x2 outputs instead of 36 on the fp32 matrix cores.
 *   dvs_wino_weights:     u [K][4][N][4] = G g G^T of every (input, output) channel pair of w [Cout][3][3][Cin];
 *                         flip = 0: K = Cin, N = Cout (forward); flip = 1: K = Cout, N = Cin, filter rotated by 180 degrees (the
 *                         data gradient of the same convolution is dvs_conv3x3_wino_fwd(dy, u_flip) with Cin/Cout swapped).
 *   dvs_wino_weights_batch: both orientations of n weights in one launch; table rows {w, u, u_flip (pointers), Cout, Cin,
 *                         first workgroup, 0 (int32)} sorted by first workgroup, ceil(Cout*Cin/256) workgroups per entry.
 *   dvs_conv3x3_wino_fwd: y [B,H,W,Cout] = [relu](conv3x3(x [B,H,W,Cin]) [+ res] + bias) (res: NULL or a tensor of y's shape -- the
 *                         skip path's gradient when the call is a BasicBlock's conv1 data gradient); stats (NULL = skip) [stat_groups][2][Cout] +=
 *                         per-channel sum / sum of squares of the raw output, as dvs_conv2d_fwd's epilogue does.  Cin % 16 == 0,
 *                         Cout % 4 == 0, tensors < 2 GiB.  as_dgrad: count the launch in the data-gradient profile slot.
 *   dvs_conv3x3_wino_gen: the same kernel behind the decoder's gathers (model/layers.py:26-41 Conv3x3 = ReflectionPad2d(1) + 3x3,
 *                         model/depth_decoder.py:52-62 upsample + concat): logical input [B,H,W,C1+C2] = cat(x [, x2]) where x is
 *                         [B,H/2,W/2,C1] when `upsample` (nearest 2x in the gather) and x2 [B,H,W,C2] the skip (NULL / C2 = 0:
 *                         none); reflect = 1: ReflectionPad2d(1), else zero padding; org = 1: y [B,H,W,Cout]; org = 2: the full
 *                         correlation y [B,H+2,W+2,Cout] (the padded-domain data gradient that dvs_reflect_fold folds back);
 *                         act: 0 none, 1 ReLU, 2 ELU.  C1 % 8 == 0, (C1+C2) % 16 == 0, Cout % 4 == 0.
 *   dvs_conv3x3_wino_wgrad: dw [Cout][3][3][Cin] += the weight gradient of that convolution from x [B,H,W,Cin] and dy [B,H,W,Cout]
 *                         (dL/dg = G^T [sum over tiles (A dY A^T) o (B^T d B)] G), the tile range split over about
 *                         target_workgroups (0 = default) workgroups that add into dw with float atomics (plain adds when
 *                         one workgroup owns a block).  Cin % 32 == 0, Cout % 32 == 0, tensors < 1 GiB (ABI 7: its offsets carry two mask bits).
 *                         dw may be shared with other launches only through float atomics: with one tile range per block (small
 *                         problems) the kernel adds with plain read-modify-writes, so no OTHER launch may touch the same dw block
 *                         concurrently (one stream per weight tensor, as the Python side keeps it). */
int dvs_wino_weights(const float* w, float* u, int Cout, int Cin, int flip, void* stream);
int dvs_conv3x3_wino_gen(const float* x, const float* x2, const float* u, const float* bias, float* y, int B, int H, int W, int C1, int C2,
                         int Cout, int Ho, int Wo, int org, int upsample, int reflect, int act, int as_dgrad, void* stream);
int dvs_conv3x3_wino_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups,
                           void* stream);
/*   dvs_conv3x3_wino_wgrad_gen (ABI 6): the same weight gradient behind the decoder's gathers (the inputs dvs_conv3x3_wino_gen reads
 *   with reflect = 1, org = 1): dw [Cout][3][3][C1+C2] += d/dw of conv3x3(ReflectionPad2d(1)(cat(x [, x2]))) from dy [B,H,W,Cout],
 *   x [B,H/2,W/2,C1] when `upsample` (nearest 2x) else [B,H,W,C1], x2 [B,H,W,C2] the skip (NULL / C2 = 0: none; only with
 *   `upsample`).  dact = 0: dy is the gradient in front of the activation (dvs_act_bwd took the derivative and the bias gradient);
 *   dact = 1 (ReLU) / 2 (ELU): dy is multiplied by act'(y_out) as it is loaded (y_out [B,H,W,Cout] = the forward output) and
 *   dbias [Cout] (NULL: none) += its column sums.  One launch per source, each adding into its channel range of dw.
 *   C1, C2, Cout % 32 == 0, H, W >= 2 (even with `upsample`), tensors < 1 GiB; the same exclusivity rule for dw as above.  Replaces the weight / bias gradient of
 *   model/layers.py:26-41 Conv3x3 (+ ELU of ConvBlock, :106-118) inside model/depth_decoder.py:52-62. */
int dvs_conv3x3_wino_wgrad_gen(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H, int W,
                               int C1, int C2, int Cout, int upsample, int dact, int target_workgroups, void* stream);
/*   Ordered weight gradient (ABI 7): the `_ws` forms take a partial-sum workspace (device, `workspace_bytes` >= what
 *   dvs_conv3x3_wino_wgrad_workspace returns for the same sizes; for the `_gen` form: the larger of its two sources, Cin = max(C1, C2))
 *   -- every workgroup stores its 32 x 32 x 9 block there and a second kernel adds the blocks into dw in split order: no float
 *   atomics on dw, bit-identical results run to run (the deterministic backward).  workspace = NULL: the atomic form above. */
size_t dvs_conv3x3_wino_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int target_workgroups);
int dvs_conv3x3_wino_wgrad_ws(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups,
                              float* workspace, size_t workspace_bytes, void* stream);
int dvs_conv3x3_wino_wgrad_gen_ws(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H,
                                  int W, int C1, int C2, int Cout, int upsample, int dact, int target_workgroups, float* workspace,
                                  size_t workspace_bytes, void* stream);
int dvs_wino_weights_batch(const void* table, int n_entries, int total_workgroups, void* stream);
int dvs_conv3x3_wino_fwd(const float* x, const float* u, const float* bias, const float* res, float* y, float* stats, int stat_groups,
                         int B, int H, int W, int Cin, int Cout, int relu, int as_dgrad, void* stream);
/*   dvs_conv3x3_wino_fwd_slots (ABI 5): the statistics go to `stat_slots` copies of the table, stats = [stat_slots][G][2][Cout]
 *   (power of two <= 64, zero-filled by the caller): workgroup w adds into copy w % stat_slots, and dvs_bn_fwd_slots /
 *   dvs_bn_finalize_slots add the copies up.  The kernel runs one workgroup per CU whose last act is these atomics; with one
 *   copy, thousands of same-address atomics serialise in the L2 and each workgroup's CU waits for them. */
int dvs_conv3x3_wino_fwd_slots(const float* x, const float* u, const float* bias, const float* res, float* y, float* stats,
                               int stat_groups, int stat_slots, int B, int H, int W, int Cin, int Cout, int relu, int as_dgrad,
                               void* stream);
/*   bf16 mode (ABI 8, dvs_set_precision(1)): the stride-1 / zero-pad-1 3x3 convolution on v_mfma_f32_32x32x16_bf16 with the input
 *   patch of a workgroup (8 x 16 output pixels + halo, 64 channels at a time) kept in LDS as bf16 -- every tap reads the same image
 *   (csrc/conv_p16.hip).  dvs_conv3x3_bf16_pack: w fp32 [Cout][3][3][Cin] -> out bf16 [9][K/16][N][16] (9 K N 2 bytes); flip = 0:
 *   K = Cin, N = Cout (forward); flip = 1: K = Cout, N = Cin, taps rotated (data gradient).  dvs_conv3x3_bf16_fwd: y [B,H,W,N] =
 *   conv3x3(x [B,H,W,K]) [+ res]; stats / stat_groups / stat_slots as in dvs_conv3x3_wino_fwd_slots (fp32 sums of the fp32 results);
 *   as_dgrad only labels the profile slot.  K % 64 == 0, N % 64 == 0, tensors < 2 GiB.  Same module as dvs_conv3x3_wino_fwd. */
/*   dvs_conv3x3_bf16_wgrad: dw [Cout][3][3][Cin] += the weight gradient of that convolution from x [B,H,W,Cin] and dy [B,H,W,Cout], bf16
 *   operands (both k-strided, fetched with ds_read_b64_tr_b16), fp32 accumulation, float atomics into dw (about target_workgroups
 *   workgroups, 0 = default; a workgroup owns a 32 x 32 x 9 block and a range of patches).  Cin % 32 == 0, Cout % 32 == 0. */
int dvs_conv3x3_bf16_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups, void* stream);
/*   dvs_conv3x3_bf16_gen: the patch kernels behind the decoder's gathers -- arguments as dvs_conv3x3_wino_gen (x / x2 / C1 / C2 / Ho / Wo /
 *   org / upsample / reflect / act), wpack from dvs_conv3x3_bf16_pack over K = C1 + C2 (its N padded to 32 columns).  C1, C2, N
 *   multiples of 64: 64-channel chunks, 64 / 128 output channels per workgroup; otherwise multiples of 16: the thin kernel (32 output
 *   channels per workgroup, the chunk's weights in LDS).  dact != 0 (thin kernel, one plain source): x is a gradient dY and y_out the
 *   forward output of the same shape, the staged operand is dY * act'(y_out) (1 ReLU, 2 ELU). */
int dvs_conv3x3_bf16_gen(const float* x, const float* x2, const void* wpack, const float* bias, float* y, int B, int H, int W, int C1, int C2,
                         int N, int Ho, int Wo, int org, int upsample, int reflect, int act, int as_dgrad, const float* y_out, int dact,
                         void* stream);
/*   dvs_conv3x3_bf16_wgrad_gen: the same behind the decoder's gathers (arguments as dvs_conv3x3_wino_wgrad_gen, + reflect): dw
 *   [Cout][3][3][C1+C2] += d/dw of conv3x3(pad(cat(x [, x2]))), dy optionally multiplied by act'(y_out) as it is staged (dact) with
 *   dbias += its column sums.  C1, C2, Cout multiples of 16 (C1 of 32 when there is a second source). */
int dvs_conv3x3_bf16_wgrad_gen(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H, int W,
                               int C1, int C2, int Cout, int upsample, int reflect, int dact, int target_workgroups, void* stream);
int dvs_conv3x3_bf16_pack(const float* w, void* out, int Cout, int Cin, int flip, void* stream);
int dvs_conv3x3_bf16_fwd(const float* x, const void* wpack, const float* res, float* y, float* stats, int stat_groups, int stat_slots, int B,
                         int H, int W, int K, int N, int as_dgrad, void* stream);
int dvs_conv2d_pack_wt(const float* w, float* wt, int Cout, int Cin, int kh, int kw, void* stream);
/*   dvs_conv2d_pack_wt_batch: the same transpose for many weights in one launch.  `table` (device memory) = n_entries
 *   records { const float* w; float* wt; int Cout, Cin, taps, wg_begin; } (32 bytes each), wg_begin = number of
 *   workgroups of the records before it, a record needs taps * ceil(Cin/32) * ceil(Cout/32) workgroups;
 *   total_workgroups = their sum.  The weights change once per optimiser step: dp.FusedAdam repacks right after it. */
int dvs_conv2d_pack_wt_batch(const void* table, int n_entries, int total_workgroups, void* stream);
/*   dvs_act_bwd: dz = dy * act'(y) over n floats (n % 4 == 0), act as in dvs_conv_fusion -- the pre-activation gradient
 *   as a tensor; dvs_conv2d_dgrad / _wgrad are then called with dact = 0 (nn.ELU / ReLU / Sigmoid backward of
 *   model/layers.py:106-118 done once instead of in both gathers).  dbias (NULL = skip): [C] += column sums of dz, the
 *   bias gradient of the [.., C] tensor (C/4 must divide 256), so that dvs_conv2d_wgrad can be called without dbias. */
int dvs_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, float* dbias, int C, void* stream);
/*   dvs_reflect_fold: second half of a reflection-padded conv's data gradient computed on the PADDED domain.  g_padded
 *   [B][H+2][W+2][C] is the gradient w.r.t. ReflectionPad2d(1)(input) -- dvs_conv2d_dgrad of the equivalent unpadded
 *   ("valid") conv on an (H+2) x (W+2) input, zero padding, which the LDS-DMA kernel serves; this folds the mirrored
 *   border back (padded row 0 -> row 1, row H+1 -> row H-2, same for columns) and, for an upsample(+concat) input
 *   (C1 > 0), sums channels < C1 over 2x2 blocks into dx [B][H/2][W/2][C1] and stores channels >= C1 to dx_skip
 *   [B][H][W][C-C1]; C1 == 0: dx [B][H][W][C].  (model/layers.py:121-136 backward, model/depthnet.py:79-85.) */
int dvs_reflect_fold(const float* g_padded, float* dx, float* dx_skip, int B, int H, int W, int C, int C1, void* stream);
/*   dx_skip / C1 (upsample+concat forward only, else NULL / 0): the gradient is split in the epilogue --
 *   channels [0,C1) are summed over each 2x2 block into dx = [B,H/2,W/2,C1] with atomics (caller zero-fills dx),
 *   channels [C1,Cin) are stored to dx_skip = [B,H,W,Cin-C1]; C1 == Cin (upsample only) needs no dx_skip. */
int dvs_conv2d_dgrad(const float* dy, const float* wt, float* dx, const dvs_conv_desc* d, const float* y_out,
                     int dact, float* dx_skip, int C1, void* stream);
/*   dvs_conv2d_dgrad_res (ABI 5): dx = data gradient + residual [B,H,W,Cin] (NULL = none; C1 must be 0).  Where a tensor feeds
 *   several consumers -- a BasicBlock input read by conv1, the 1x1 downsample branch and, in DepthNet, the decoder's skip
 *   connection (torchvision BasicBlock through model/resnet_encoder.py:94-111, model/depthnet.py:79-85) -- autograd would add
 *   the consumers' gradients in one more pass over the tensor each; the gradient already computed is added in this kernel's
 *   epilogue instead.  Same for dvs_conv2d_head_bwd_res (dx += dx_residual) and dvs_maxpool3x3s2_bwd_res below. */
int dvs_conv2d_dgrad_res(const float* dy, const float* wt, float* dx, const dvs_conv_desc* d, const float* y_out,
                         int dact, float* dx_skip, int C1, const float* residual, void* stream);
int dvs_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, const dvs_conv_desc* d,
                     const dvs_conv_fusion* f, const float* y_out, int dact, void* stream);
/*   Ordered form (ABI 7): with a slab workspace of >= dvs_conv2d_wgrad_workspace(...) bytes every workgroup of the split-K implicit
 *   GEMM stores its partial tile with plain stores and a second kernel adds the splits into dw in split order -- no float atomics
 *   on dw (they run at a fifth of the plain-store rate on this chip and make the result depend on timing).  The stem and the thin
 *   full-resolution decoder layers have no slab path (workspace size 0: the call behaves like dvs_conv2d_wgrad); a bias gradient
 *   (dbias) still rides on atomics.  workspace = NULL: dvs_conv2d_wgrad. */
size_t dvs_conv2d_wgrad_workspace(const dvs_conv_desc* d, const dvs_conv_fusion* f, int dact, int with_bias);
int dvs_conv2d_wgrad_ws(const float* x, const float* dy, float* dw, float* dbias, const dvs_conv_desc* d, const dvs_conv_fusion* f,
                        const float* y_out, int dact, float* workspace, size_t workspace_bytes, void* stream);

/* Narrow output heads (Cout in {1,2,6,8}, Cin % 4 == 0, stride 1, "same" size: 2*pad == k-1) on the vector
 * ALUs: DepthNet's dispconv layers (model/depthnet.py:57-58,86-88) and PoseNet's last 1x1
 * (model/posenet_single.py:165).  `act` as in dvs_conv_fusion.  bwd: dx (NULL = skip), dw/dbias accumulated
 * with atomics (caller zero-fills); y / dy are the forward output and its gradient. */
int dvs_conv2d_head_fwd(const float* x, const float* w, const float* bias, float* y, const dvs_conv_desc* d, int act,
                        void* stream);
int dvs_conv2d_head_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                        float* dbias, const dvs_conv_desc* d, int act, void* stream);
int dvs_conv2d_head_bwd_res(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                            float* dbias, const dvs_conv_desc* d, int act, const float* dx_residual, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1  the ResNet stem's MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC tensors
 *     (torchvision resnet18.maxpool reached through model/resnet_encoder.py:104).
 *     x, dx: [B,H,W,C]; y, dy: [B,Ho,Wo,C] with Ho = (H-1)/2+1; idx: [B,Ho,Wo,C] bytes, the winning tap
 *     ky*3+kx (first maximum in scan order, as torch); C % 4 == 0.  bwd writes every element of dx.
 * ------------------------------------------------------------------------------------------- */
int dvs_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, void* stream);
int dvs_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, void* stream);
/* dx = maxpool gradient + residual [B,H,W,C] (NULL = none): the stem output feeds the pool and DepthNet's finest skip. */
int dvs_maxpool3x3s2_bwd_res(const float* dy, const unsigned char* idx, float* dx, const float* residual, int B, int H, int W,
                             int C, void* stream);
/* `upsample` of model/layers.py:196-199 (F.interpolate(scale_factor=2, mode="nearest")) as a standalone operator on NHWC
 * tensors: x, dx [B,H,W,C]; y, dy [B,2H,2W,C]; C % 4 == 0.  The backward is the 2x2 block sum.  (The decoder itself never
 * materialises the upsampled tensor: dvs_conv_fusion.x2 / C1 gather it inside the convolution.) */
int dvs_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int dvs_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1  training-mode BatchNorm2d (+ residual add, + ReLU) of the ResNet BasicBlocks on NHWC tensors
 *     (torchvision BasicBlock tail used by model/resnet_encoder.py:100-111).  `stats` = [2][C] per-channel
 *     sum / sum of squares of y as accumulated by dvs_conv2d_fwd's epilogue; count = B*H*W.
 *       finalize : mean, invstd = rsqrt(biased var + eps); scale = gamma*invstd; shift = beta - mean*scale;
 *                  running_mean/var updated with `momentum` (unbiased var), NULL = leave them;
 *                  *num_batches_tracked (int64, NULL = skip) is incremented, as nn.BatchNorm2d does.
 *       apply_fwd: z = [relu](y*scale + shift [+ residual | + residual*res_scale + res_shift]).
 *       bwd_reduce: du = dz * [z > 0] (z NULL = no ReLU; du NULL = do not store); sums[0][c] += sum du,
 *                  sums[1][c] += sum du * xhat (= d beta, d gamma); caller zero-fills sums.  Each workgroup
 *                  writes one partial row into `workspace` (dvs_bn_bwd_workspace bytes), a second small
 *                  kernel adds the rows into sums (same-address float atomics serialise at ~85 ns each).
 *       bwd_apply: dy = gamma * invstd * (du - sums0/M - xhat * sums1/M); dgamma_acc / dbeta_acc (both or
 *                  neither, NULL = skip): d gamma += sums1, d beta += sums0, i.e. the parameter gradients are
 *                  accumulated in place (what autograd's AccumulateGrad would do with one more launch each).
 *     M = pixels (rows), C % 4 == 0 and C/4 divides 256 (C in {16..1024}).
 *     groups G >= 1 (PoseNet evaluates its two frame pairs as one batch, each half normalised on its own): ONE launch
 *     serves G independent sub-batches of M rows each, stored back to back -- tensors [G*M][C], stats / sums [G][2][C],
 *     scale / shift / mean / invstd = rows of a [G][4][C] table (pass the group-0 rows); gamma / beta / running
 *     statistics / gradient sinks are shared (running statistics get the G updates in order).
 * ------------------------------------------------------------------------------------------- */
int dvs_bn_finalize(const float* stats, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                    float* invstd, int C, long long* num_batches_tracked, int groups, void* stream);
/* finalize + apply_fwd in ONE launch (what the training forward uses): `stats` [G][2][C] raw sums from the conv epilogue,
 * `fin` [G][4][C] out (scale, shift, mean, invstd: kept for the backward); residual / res_scale / res_shift as in
 * dvs_bn_apply_fwd (res_scale / res_shift = group-0 rows of the downsample branch's own [G][4][C] table, produced by
 * dvs_bn_finalize). */
int dvs_bn_fwd(const float* y, const float* stats, double count, const float* gamma, const float* beta,
               float* running_mean, float* running_var, float momentum, float eps, long long* num_batches_tracked,
               float* fin, const float* residual, const float* res_scale, const float* res_shift, float* z, size_t M,
               int C, int relu, int groups, void* stream);
/* the same with `stats` = [stat_slots][G][2][C]: copies that are added up on load (dvs_conv3x3_wino_fwd_slots) */
int dvs_bn_fwd_slots(const float* y, const float* stats, int stat_slots, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, long long* num_batches_tracked,
                     float* fin, const float* residual, const float* res_scale, const float* res_shift, float* z, size_t M,
                     int C, int relu, int groups, void* stream);
int dvs_bn_finalize_slots(const float* stats, int stat_slots, double count, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                          float* invstd, int C, long long* num_batches_tracked, int groups, void* stream);
int dvs_bn_apply_fwd(const float* y, const float* scale, const float* shift, const float* residual,
                     const float* res_scale, const float* res_shift, float* z, size_t M, int C, int relu, int groups,
                     void* stream);
/* bytes of the per-workgroup partial-sum workspace of dvs_bn_bwd_reduce (0 = unsupported shape) */
size_t dvs_bn_bwd_workspace(size_t M, int C, int groups);
int dvs_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                      float* du, float* sums, float* workspace, size_t M, int C, int groups, void* stream);
int dvs_bn_bwd_apply(const float* du, const float* y, const float* mean, const float* invstd, const float* gamma,
                     const float* sums, float* dy, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups,
                     void* stream);
/* Stem tail (ABI 5): relu(bn1(conv1(x))) -> MaxPool2d(3, 2, 1) of model/resnet_encoder.py:102-104 without the tensor in between.
 *   fwd: y [B,H,W,C] = raw conv1 output, fin [G][4][C] = scale, shift, mean, invstd from dvs_bn_finalize (group = image / (B/G));
 *        pooled [B,Ho,Wo,C] and idx (one byte per element, winning tap ky*3+kx, first maximum in scan order) as
 *        dvs_maxpool3x3s2_fwd would give on z = relu(y*scale+shift); z [B,H,W,C] is written only when non-NULL (DepthNet's
 *        finest skip connection reads it; PoseNet never does).
 *   bwd: dz = maxpool gradient (gathered from dpool / idx, never stored) + dz_extra (NULL or [B,H,W,C]: the gradient of z's
 *        other consumer), ReLU mask recomputed from y; sums [G][2][C] receive sum(dz), sum(dz*xhat);
 *        dy [B,H,W,C] = training-mode BatchNorm backward; dgamma_acc / dbeta_acc as in dvs_bn_bwd_apply.  workspace:
 *        dvs_bn_bwd_slot_floats(C, G) ZERO-FILLED floats (the slot table of dvs_bn_bwd); sums [G][2][C] is written, not
 *        accumulated.  C/4 must divide 256. */
/* (ABI 5) reduce + apply without the kernel in between that adds the per-workgroup partial rows: `slot_table` =
 *   dvs_bn_bwd_slot_floats(C, groups) zero-filled floats ([G][32][2][C]); workgroup w of the reduce pass adds its partial sums into
 *   copy w % 32 and every workgroup of the apply pass adds the copies up.  fin = the forward's [G][4][C] table.  ymask != 0: ReLU
 *   without a residual, the mask recomputed from y (z and du must be NULL); else z (NULL = no ReLU) masks dz and du (NULL = not
 *   needed) receives dz * [z > 0].  sums_out (NULL = skip): [G][2][C] = sum(du), sum(du * xhat) for callers that hand d beta /
 *   d gamma to autograd.  The deterministic mode (dvs_set_deterministic) keeps dvs_bn_bwd_reduce / _apply with their ordered sum. */
int dvs_bn_bwd_slot_floats(int C, int groups);
int dvs_bn_bwd(const float* dz, const float* z, const float* y, const float* fin, int ymask, const float* gamma, float* du, float* dy,
               float* slot_table, float* sums_out, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups, void* stream);
int dvs_bn_relu_maxpool_fwd(const float* y, const float* fin, float* z, float* pooled, unsigned char* idx, int B, int H, int W, int C,
                            int groups, void* stream);
int dvs_bn_relu_maxpool_bwd(const float* dpool, const unsigned char* idx, const float* dz_extra, const float* y, const float* fin,
                            const float* gamma, float* sums, float* workspace, float* dy, int B, int H, int W, int C,
                            float* dgamma_acc, float* dbeta_acc, int groups, void* stream);
/* ReLU without a residual (bn1 of a BasicBlock, the stem): the mask z > 0 is recomputed from y with the forward's own
 * expression max(y*scale + shift, 0) (scale / shift = rows 0 / 1 of dvs_bn_fwd's `fin` table), so the backward neither reads z
 * nor writes / re-reads du: 5 tensor passes instead of 7.  `dz` takes du's place in the apply call. */
int dvs_bn_bwd_reduce_ymask(const float* dz, const float* y, const float* mean, const float* invstd, const float* scale,
                            const float* shift, float* sums, float* workspace, size_t M, int C, int groups, void* stream);
int dvs_bn_bwd_apply_ymask(const float* dz, const float* y, const float* mean, const float* invstd, const float* scale,
                           const float* shift, const float* gamma, const float* sums, float* dy, size_t M, int C, float* dgamma_acc,
                           float* dbeta_acc, int groups, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a4  axis-angle + translation -> 4x4 camera motion
 *     replaces transformation_from_parameters / rot_from_axisangle / get_translation_matrix
 *     (vo/learner_func.py:29-104 == model/layers.py:28-103).
 *     axisangle, translation: [B,3]; M, dM: [B,4,4]; invert as in the reference (R^T . T(-t)).
 * ------------------------------------------------------------------------------------------- */
int dvs_pose_to_mat_fwd(const float* axisangle, const float* translation, int invert,
                        float* M, int B, void* stream);
int dvs_pose_to_mat_bwd(const float* axisangle, const float* translation, int invert,
                        const float* dM, float* d_axisangle, float* d_translation,
                        int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a5-a12  fused view-synthesis loss chain
 *     replaces MonodepthTrainer._generate_images_pred + _compute_losses
 *     (vo/learner_new.py:132-258) and the operators they call: F.interpolate bilinear
 *     (learner_new.py:136-140), disp_to_depth (learner_func.py:16-26), BackprojectDepth
 *     (:106-135), Project3D (:137-159), F.grid_sample border/align_corners (learner_new.py:165-170),
 *     SSIM (learner_func.py:177-207), _compute_reprojection_loss (learner_new.py:60-74), auto-mask
 *     min (learner_new.py:212-244), disparity normalisation + get_smooth_loss
 *     (learner_new.py:246-252, learner_func.py:161-174).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int B, H, W;
    int num_scales;          /* 1..4; scale s has a [B,1,hs[s],ws[s]] disparity map */
    int hs[DVS_MAX_SCALES];
    int ws[DVS_MAX_SCALES];
    int auto_mask;           /* learner_new.py:37,212-231 */
    float min_depth, max_depth;      /* learner_new.py:39-40 */
    float ssim_ratio;                /* learner_new.py:38 */
    float smoothness_ratio;          /* learner_new.py:36 */
} dvs_chain_cfg;

typedef struct {
    /* inputs */
    const float* target;             /* [B,3,H,W]  sample[("target_image",0)] */
    const float* source[2];          /* [B,3,H,W]  0: ("source_left",0) frame -1, 1: ("source_right",0) frame +1 */
    const float* disp[DVS_MAX_SCALES]; /* [B,1,hs,ws] outputs[("disp",s)] */
    const float* K;                  /* [B,4,4]    sample[("K",0)] */
    const float* inv_K;              /* [B,4,4]    sample[("inv_K",0)] */
    const float* T[2];               /* [B,4,4]    outputs[("cam_T_cam",0,-1/+1)] */
    const float* noise;              /* [S,B,2,H,W] standard-normal tie-break noise (x1e-5 applied
                                        inside), or NULL: counter-based Philox noise from `seed` */
    uint64_t seed;
    /* workspace (sizes from dvs_chain_workspace) */
    float* partials;                 /* per-block partial sums */
    uint8_t* sel;                    /* [B,H,W] argmin of the 4-way min, 2 bits per scale */
    float* stats;                    /* [B,S,4] per-image sums {min-loss, disp_up, Gx, Gy}, followed by a per-image camera
                                        table (inv_K[:3,:3], (K.T_-1)[:3,:], (K.T_+1)[:3,:]) and RGBA-packed copies of the two
                                        source frames (2*B*H*W*16 bytes) that the forward call writes and the backward call
                                        reads: allocate dvs_chain_workspace's stats_bytes and keep it until the backward */
    /* outputs */
    float* losses;                   /* [S] losses["loss/s"] */
    /* optional materialised tensors of the reference's `outputs` dict (NULL = skip) */
    float* disp_up[DVS_MAX_SCALES];  /* [B,1,H,W]   ("disp_up",s) */
    float* depth[DVS_MAX_SCALES];    /* [B,1,H,W]   ("depth",s) */
    float* grid[DVS_MAX_SCALES][2];  /* [B,H,W,2]   ("sample",f,s) */
    float* color[DVS_MAX_SCALES][2]; /* [B,3,H,W]   ("color",f,s) */
} dvs_chain_fwd_io;

typedef struct {
    const float* d_losses;           /* [S] upstream gradient of losses["loss/s"] */
    float* d_disp[DVS_MAX_SCALES];   /* [B,1,hs,ws]; overwritten (zero-filled inside where needed) */
    float* d_T[2];                   /* [B,4,4] gradient wrt cam_T_cam(-1/+1) */
    float* bwd_partials;             /* workspace */
    int scale_begin, scale_end;      /* scales [begin, end) handled by this call; 0, 0 = all of them */
    int phase;                       /* 0: gradient kernel + d_T reduction; 1: gradient kernel only; 2: d_T reduction only
                                        (after calls with phase 1 have covered every scale) -- lets a caller launch the
                                        scales on different streams and hand d disp_0 to the decoder's backward first */
} dvs_chain_bwd_io;

/* Bytes of each workspace buffer for a configuration (host out-params). */
int dvs_chain_workspace(const dvs_chain_cfg* cfg, size_t* partials_bytes, size_t* sel_bytes,
                        size_t* stats_bytes, size_t* bwd_partials_bytes);
int dvs_chain_fwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, void* stream);
/* Needs the same inputs/workspace as the forward call it differentiates (sel, stats filled). */
int dvs_chain_bwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, const dvs_chain_bwd_io* g,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Standalone (un-fused) operators for callers that use model/layers.py piecewise
 * ------------------------------------------------------------------------------------------- */
/* BackprojectDepth.forward (vo/learner_func.py:130-135): depth [B,1,H,W], inv_K [B,4,4] ->
 * cam_points [B,4,H*W] (row 3 = 1).  bwd: d_depth = sum_{r<3} d_cam[r] * (inv_K[:3,:3].[x,y,1])[r]. */
int dvs_backproject_fwd(const float* depth, const float* inv_K, float* cam_points, int B, int H, int W,
                        void* stream);
int dvs_backproject_bwd(const float* d_cam_points, const float* inv_K, float* d_depth, int B, int H,
                        int W, void* stream);

/* Project3D.forward (vo/learner_func.py:148-159): points [B,4,H*W], K,T [B,4,4] -> grid [B,H,W,2]
 * normalised to [-1,1].  bwd: d_points [B,4,H*W], d_T [B,4,4]; workspace bytes from
 * dvs_project_bwd_workspace. */
int dvs_project_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H,
                    int W, float eps, void* stream);
size_t dvs_project_bwd_workspace(int B, int H, int W);
int dvs_project_bwd(const float* points, const float* K, const float* T, const float* d_grid,
                    float* d_points, float* d_T, float* workspace, int B, int H, int W, float eps,
                    void* stream);

/* SSIM.forward (vo/learner_func.py:193-207) on `planes` = B*C independent [H,W] planes:
 * clamp((1 - SSIM)/2, 0, 1) with ReflectionPad2d(1) + AvgPool2d(3,1).  bwd: d_x and/or d_y (either
 * may be NULL). */
int dvs_ssim_fwd(const float* x, const float* y, float* out, int planes, int H, int W, void* stream);
int dvs_ssim_bwd(const float* x, const float* y, const float* d_out, float* d_x, float* d_y,
                 int planes, int H, int W, void* stream);

/* get_smooth_loss (vo/learner_func.py:161-174): disp [B,1,H,W], img [B,C,H,W] -> out[1] (device
 * scalar).  bwd: d_disp = d_out[0] * d out / d disp (the image is data). */
size_t dvs_smooth_workspace(int B, int H, int W);
int dvs_smooth_fwd(const float* disp, const float* img, float* out, float* workspace, int B, int C,
                   int H, int W, void* stream);
int dvs_smooth_bwd(const float* disp, const float* img, const float* d_out, float* d_disp, int B,
                   int C, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * the supervised depth learner's multi-scale loss, depth/depth_learner.py:51-117
 *     (DepthLearner.multi_scale_loss: F.interpolate bilinear of each scale's depth to (H, W), get_smooth_loss on the
 *     mean-normalised map, silog_loss over the valid pixels).  One forward launch for all scales, one backward launch.
 *       pred_depth[s] [B,1,hs,ws] (disp_to_depth already applied, as the reference's pred_depths list),
 *       gt_depth [B,1,H,W], valid_mask [B,1,H,W] bytes (non-zero = valid), rgb [B,3,H,W];
 *       out [2*S] (device): silog_s for s < S, then smooth_s;  d_out [2*S]: their upstream gradients;
 *       d_pred_depth[s] [B,1,hs,ws], overwritten.  workspace: dvs_depth_loss_workspace bytes, filled by the forward
 *       call and read by the backward call.  No valid pixel at all gives NaN, as the reference's mean over nothing does.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int B, H, W;
    int num_scales;
    int hs[DVS_MAX_SCALES];
    int ws[DVS_MAX_SCALES];
    float variance_focus;            /* 0.85, depth_learner.py:80 */
} dvs_depth_loss_cfg;
size_t dvs_depth_loss_workspace(const dvs_depth_loss_cfg* cfg);
int dvs_depth_loss_fwd(const dvs_depth_loss_cfg* cfg, const float* const* pred_depth, const float* gt_depth,
                       const unsigned char* valid_mask, const float* rgb, float* workspace, float* out, void* stream);
int dvs_depth_loss_bwd(const dvs_depth_loss_cfg* cfg, const float* const* pred_depth, const float* gt_depth,
                       const unsigned char* valid_mask, const float* rgb, float* workspace, const float* d_out,
                       float* const* d_pred_depth, void* stream);

/* ---------------------------------------------------------------------------------------------
 * the input side of the path, vo/dataset/common.py:38-92, behind the H2D copy
 *   dvs_u8_to_f32_planar: src uint8 [N,H,W,3] (RGB, or BGR with bgr != 0) -> dst fp32 [N,3,H,W] (RGB planes) / 255:
 *       transforms.ToTensor of common.py:77; with bgr the preprocessing of slam/network.py:42-50.  H*W % 4 == 0,
 *       src 4-byte and dst 16-byte aligned.
 *   dvs_color_jitter: torchvision ColorJitter (common.py:31-37,79-81) in place on images fp32 [N,3,H,W] in [0,1].
 *       records (device): N x { int order[4]; float factor[4]; } -- order = adjustment ids in application order
 *       (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none), factor[id] = that adjustment's factor (hue: the shift
 *       in [-0.5, 0.5]).  Images that share a record's values get the same jitter (the reference jitters the three frames
 *       of a sample together); the contrast step's mean gray level is per image.  workspace: dvs_color_jitter_workspace
 *       bytes.
 *   dvs_resample_u8: one pass of PIL's Image.resize(size, Image.BILINEAR) on uint8 frames [N,h,w,3] (common.py:38-44:
 *       antialiased triangle filter, 22-bit integer coefficients, uint8 rounding after each pass) -- bit exact.  axis 0
 *       resamples along x ([N,h,in_w,3] -> [N,h,out_w,3]), axis 1 along y; bounds [out_n][2] = (first input index, tap
 *       count), coef [out_n][ksize] integer taps, both built on the host exactly as Pillow does
 *       (input_pipeline.pil_bilinear_tables).  A full resize = the x pass, then the y pass.
 */
int dvs_resample_u8(const unsigned char* src, unsigned char* dst, const int* bounds, const int* coef, int ksize, int N, int in_h,
                    int in_w, int out_h, int out_w, int axis, void* stream);
int dvs_u8_to_f32_planar(const unsigned char* src, float* dst, int N, int H, int W, int bgr, void* stream);
size_t dvs_color_jitter_workspace(int N, int H, int W);
int dvs_color_jitter(float* images, const void* records, float* workspace, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a14 / SURVEY.md 8(f) rank 1: Depth-Anything-V2 ViT-S (BASELINE.json configs[4]) -- forward kernels beside the
 *     implicit-GEMM engine (the token GEMMs are dvs_conv2d_fwd calls on [1,1,M,K] tensors).
 *   dvs_attention_fwd: out [B,N,heads*64] = softmax(q k^T * scale) v per head from qkv [B,N,3,heads,64] (the output of the
 *       qkv Linear; lse may be NULL), model/depth_anything_v2/dinov2_layers/attention.py:49-62; flash style on the fp32 matrix cores.
 *   dvs_layernorm_fwd: nn.LayerNorm(C, eps) over the last dimension of x [M,C] (block.py:53,67; dinov2.py:166).
 *   dvs_vit_patchify: image [B,3,H,W] -> rows [B*(H/p)*(W/p)][k_padded], row = one patch in (ci, ky, kx) order, zero padded
 *       to k_padded (>= 3 p p, % 4 == 0): the A operand of PatchEmbed.proj (patch_embed.py:69-82).
 *   dvs_vit_assemble: x [B,Np+1,C] = cat(cls_token, patch_tokens [B,Np,C]) + pos_embed [Np+1,C] (dinov2.py:219-229).
 *   dvs_resize_bilinear_ac: F.interpolate(mode="bilinear", align_corners=True) of an NHWC map [B,h,w,C] to [B,H,W,C]
 *       (util/blocks.py:143, dpt.py:145).
 *   dvs_deconv_shuffle: y [B,h*k,w*k,Cout] from g [B,h,w,k*k*Cout] = the 1x1 product of the input with the ConvTranspose2d
 *       weight arranged [(a*k+c)*Cout+co][ci]: nn.ConvTranspose2d(kernel_size=k, stride=k) of dpt.py:60-73.
 * ------------------------------------------------------------------------------------------- */
int dvs_attention_fwd(const float* qkv, float* out, float* lse, int B, int N, int heads, int head_dim, float scale, void* stream);
int dvs_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int C, float eps, void* stream);
/* Backward halves (training of the encoder swap).  lse [B,heads,N]: log-sum-exp of every score row, written by
 * dvs_attention_fwd when non-NULL.  dvs_attention_bwd: d_qkv [B,N,3,heads,64] from d_out [B,N,heads*64] (delta [B,heads,N] is
 * scratch); two recomputing kernels in the forward's register layout, no atomics.  dvs_layernorm_bwd: dx, and
 * dgamma_acc / dbeta_acc += (caller zero-fills or accumulates).  dvs_act_fwd / dvs_act_bwd_in: ReLU (1) / GELU (4) as
 * their own passes, the derivative taken from the activation's INPUT.  dvs_resize_bilinear_ac_bwd: dx [B,h,w,C] (zero-filled
 * inside) from dy [B,H,W,C].  dvs_deconv_unshuffle: the inverse permutation of dvs_deconv_shuffle. */
int dvs_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, float* delta, float* d_qkv, int B,
                      int N, int heads, int head_dim, float scale, void* stream);
int dvs_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma_acc, float* dbeta_acc, int M, int C,
                      float eps, void* stream);
int dvs_act_fwd(const float* x, float* y, size_t n, int act, void* stream);
int dvs_act_bwd_in(const float* x, const float* dy, float* dx, size_t n, int act, void* stream);
int dvs_resize_bilinear_ac_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, int C, void* stream);
int dvs_deconv_unshuffle(const float* dy, float* dg, int B, int h, int w, int k, int Cout, void* stream);
int dvs_vit_patchify(const float* image, float* rows, int B, int H, int W, int patch, int k_padded, void* stream);
int dvs_vit_assemble(const float* patch_tokens, const float* cls_token, const float* pos_embed, float* x, int B, int num_patches,
                     int C, void* stream);
int dvs_resize_bilinear_ac(const float* x, float* y, int B, int h, int w, int H, int W, int C, void* stream);
int dvs_deconv_shuffle(const float* g, float* y, int B, int h, int w, int k, int Cout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_H */
