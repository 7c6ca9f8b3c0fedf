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
    float* stats;                    /* [B,S,4] per-image {mean_disp, Gx, Gy, unused} */
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
    float* d_disp[DVS_MAX_SCALES];   /* [B,1,hs,ws]; must be zero-filled by the caller for s>0 */
    float* d_T[2];                   /* [B,4,4] gradient wrt cam_T_cam(-1/+1) */
    float* bwd_partials;             /* workspace */
} dvs_chain_bwd_io;

/* Bytes of each workspace buffer for a configuration (host out-params). */
int dvs_chain_workspace(const dvs_chain_cfg* cfg, size_t* partials_bytes, size_t* sel_bytes,
                        size_t* stats_bytes, size_t* bwd_partials_bytes);
int dvs_chain_fwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, void* stream);
/* Needs the same inputs/workspace as the forward call it differentiates (sel, stats filled). */
int dvs_chain_bwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, const dvs_chain_bwd_io* g,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_H */
