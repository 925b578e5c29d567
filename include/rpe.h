/*
 * rpe.h -- C ABI of librpe_hip.so: the MI355X (gfx950) hot path of the per-frame stereo pose solve.
 *
 * Drop-in boundary for aimi-lab/robust-pose-estimator (reference paths are relative to its repo root).
 * Every entry point takes raw DEVICE pointers, explicit shapes and a HIP stream (passed as void*, a
 * hipStream_t; NULL = the default stream).  All tensors are caller-owned, contiguous, NCHW, row-major,
 * x fastest.  Inputs are const; outputs and scratch are pre-allocated by the caller (size-query
 * functions below).  The library keeps no mutable global state and is re-entrant (one host thread per
 * GPU).  Return value: 0 = ok; <0 = RPE_E_* below.  No C++ exception crosses this boundary.  Numerical
 * failure is signalled in-band exactly like the reference (NaN pose -> caller's failure gate,
 * core/pose/pose_estimator.py:81-87).
 */
#ifndef RPE_H
#define RPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPE_OK 0
#define RPE_E_BADARG (-1)  /* null pointer, non-positive size, unsupported dtype / radius / level count */
#define RPE_E_LAUNCH (-2)  /* hipGetLastError() != hipSuccess after a launch                            */
#define RPE_E_UNSUPPORTED (-3)

#define RPE_F32 0
#define RPE_F64 1
#define RPE_F16 2      /* rpe_corr_build_ex only: feature maps rounded to fp16 */
#define RPE_F32X3 3    /* rpe_corr_build_ex only: f32 feature maps, every f32 product evaluated as six bf16 products (3-way exact split) */

/* solver modes of rpe_pose_solve */
#define RPE_SOLVER_LBFGS 0 /* reference-faithful: torch.optim.LBFGS(lr=1, line_search_fn=None) iterates */
#define RPE_SOLVER_GN 1    /* Gauss-Newton: 6x6 normal equations, Cholesky in f64                        */

/* stop reasons written to info[.][2] by rpe_pose_solve (0 = still running / not started) */
#define RPE_STOP_OPT_AT_START 1 /* max|g| <= 1e-7 at the first evaluation                                */
#define RPE_STOP_GTD 2          /* directional derivative g.d > -1e-9                                    */
#define RPE_STOP_MAX_ITER 3
#define RPE_STOP_MAX_EVAL 4     /* evals >= max_iter*5/4                                                 */
#define RPE_STOP_OPT 5          /* max|g| <= 1e-7                                                        */
#define RPE_STOP_STEP 6         /* max|t*d| <= 1e-9                                                      */
#define RPE_STOP_LOSS 7         /* |loss - prev_loss| < 1e-9                                             */
#define RPE_STOP_NOT_PD 8       /* GN only: H not positive definite or g not finite; pose left unchanged */

/* Library / build identification, e.g. "rpe-hip 0.1 gfx950". */
const char *rpe_version(void);
/* Layout version of this header's structs (rpe_conv_desc, rpe_solve_opts) and signatures: a binding compares it with the
 * RPE_ABI_VERSION it was written against before the first call (the ctypes binding does, robust-pose-estimator_amd/_lib.py).
 * 5: rpe_conv_desc is 200 bytes (stats_tiles), stride-2 statistics are one record per 32 output pixels, rpe_pose_solve_ex, rpe_pose_gate_chain.
 * Entry points are only ever ADDED under one RPE_ABI_VERSION (they change no struct and no existing signature); RPE_ABI_MINOR counts those
 * additions, so a binding can require "version 5, minor >= m" for the newest entry point it calls -- or probe with dlsym:
 *   minor 0: the 68 entry points of round 4;  1: rpe_conv_wino_x3*, rpe_conv1x1_x3*, rpe_conv_wino1d_x3* (9, round 5);  2: rpe_run_ops and
 *   the rpe_*_args structs of the prepared launch lists, rpe_corr_lookup_conv1x1* (round 6). */
#define RPE_ABI_VERSION 5
#define RPE_ABI_MINOR 2
int rpe_abi_minor(void);
int rpe_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * SE(3) group ops -- replace the lietorch C++/CUDA kernels the reference calls through
 * `from lietorch import SE3` (core/geometry/pinhole_transforms.py:3,29,51; core/pose/pose_estimator.py:81-91;
 * core/optimization/declerative_node_lie.py:233-234).  Pose = 7 scalars [tx ty tz qx qy qz qw]; tangent =
 * 6 scalars [tau(3) phi(3)].  dtype = RPE_F32 | RPE_F64 for every pointer of the call.
 * --------------------------------------------------------------------------------------------------------- */
int rpe_se3_exp(const void *xi, void *T, int64_t n, int dtype, void *stream);            /* (n,6) -> (n,7)   */
int rpe_se3_log(const void *T, void *xi, int64_t n, int dtype, void *stream);            /* (n,7) -> (n,6)   */
int rpe_se3_mul(const void *A, const void *B, void *C, int64_t n, int dtype, void *stream); /* C = A * B     */
int rpe_se3_inv(const void *T, void *Tinv, int64_t n, int dtype, void *stream);
/* T (n,7) acts on pts (n,m,3) -> out (n,m,3): `T * pts` with the (n,1) pose broadcast over m points.      */
int rpe_se3_act(const void *T, const void *pts, void *out, int64_t n, int64_t m, int dtype, void *stream);
/* Trajectory chaining of core/pose/pose_estimator.py:90-91 as an inclusive scan over m relative poses:
 * P_k = P_{k-1} * inv(scale(rel_k, s)), P_{-1} = init (7 scalars, may be NULL = identity). f64 or f32.    */
int rpe_se3_chain(const void *rel, const void *init, void *out, int64_t m, double scale, int dtype, void *stream);
/* The whole per-frame bookkeeping of PoseEstimator.forward (core/pose/pose_estimator.py:81-91) for m consecutive relative poses:
 * row k fails when any of its 7 scalars is NaN or any |log(rel_k)| > thr (:81) and is replaced by the identity (:83); then the chain
 * of rpe_se3_chain.  rel_out (m,7) = the gated relative poses (may be NULL), abs_out (m,7) = the chained absolute poses, ok (m) int32 =
 * 1 where the row passed (may be NULL).  Same arithmetic as rpe_se3_log / _inv / _mul step by step: bit-identical poses.            */
int rpe_pose_gate_chain(const void *rel, const void *init, void *rel_out, void *abs_out, int32_t *ok, int64_t m, double scale,
                        double thr, int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Pose layer -- replaces DPoseSE3Head.objective / .solve (core/pose/pose_head.py:12-79) together with
 * torch.optim.LBFGS, transform/project (core/geometry/pinhole_transforms.py:28-30,90-99) and
 * DeclarativeFunctionLie.forward (core/optimization/declerative_node_lie.py:223-247).
 *
 * Inputs (all float32 unless noted), n rows, H*W pixels each:
 *   flow (n,2,H,W)  pcl1 (n,3,H,W)  pcl2 (n,3,H,W)  w1 (n,1,H,W)  w2 (n,1,H,W)
 *   mask1, mask2 (n,1,H,W) uint8 (torch.bool storage, 0/1)   K (n,3,3)   loss_weight (n,2) = [w3d, w2d]
 * All arithmetic is float64, as in the reference (pose_head.py:64).
 * --------------------------------------------------------------------------------------------------------- */
/* Scratch bytes needed by rpe_pose_reduce / rpe_pose_solve for n rows of h*w pixels. */
size_t rpe_pose_workspace_bytes(int n, int h, int w);

/* One objective evaluation at poses T (n,7) f64.  out (n,32) f64 per row:
 *   [0] loss2d  [1] loss3d  [2] f = lw[1]*loss2d + lw[0]*loss3d  [3..8] g = df/dxi (left perturbation,
 *   unclipped)  [9..29] upper triangle of the Gauss-Newton Hessian H (row-major: 00 01 .. 05 11 12 .. 55;
 *   zeros unless need_hessian != 0)  [30],[31] reserved.                                                  */
int rpe_pose_reduce(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                    const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                    const double *T, int n, int h, int w, int need_hessian, double *out, void *workspace,
                    void *stream);

/* Whole solve on the device, no host synchronisation: start at identity, run `iters` iterations of
 * `mode`, n independent rows.  Outputs: T_out (n,7) f64 group element; vec7 (n,7) f32 and log6 (n,6) f32
 * (the two tensors DeclarativeFunctionLie.forward returns); info (n,4) int32 = [n_iter, func_evals,
 * stop_reason, 0].  Any output pointer except T_out may be NULL.                                          */
int rpe_pose_solve(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                   const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                   int n, int h, int w, int mode, int iters, double *T_out, float *vec7, float *log6,
                   int32_t *info, void *workspace, void *stream);

/* Same with torch.optim.LBFGS's remaining constructor arguments exposed (rpe_pose_solve uses its defaults
 * tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100, which is how pose_head.py:70 builds the optimiser).
 * 1 <= history_size <= 100.  The tolerances also drive the Gauss-Newton step test. */
int rpe_pose_solve_opts(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                        const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                        int n, int h, int w, int mode, int iters, double tolerance_grad, double tolerance_change,
                        int history_size, double *T_out, float *vec7, float *log6, int32_t *info, void *workspace,
                        void *stream);

/* The same with the options in a sized struct: struct_size = sizeof(rpe_solve_opts) of the header the caller was built against.  The library
 * accepts any struct_size >= the layout it knows (fields are only ever appended, so a newer caller's longer struct carries this layout as
 * its prefix) and returns RPE_E_BADARG for a shorter one; a field appended later is read only when struct_size covers it and takes its
 * documented default otherwise.
 *   partition_rows: the pixel reduction sums per-block partials in a fixed order, and the number of blocks per row is chosen so that
 *   ONE round of workgroups covers the batch -- so a row's float64 sums are grouped differently in a 16-row launch than alone.
 *   0 = that default; p > 0 = the partition a p-row batch would get.  With p = 1 every row's iterates are bit-identical to solving
 *   it alone, whatever the batch (what the chunked sequence tracker asks for: core/pose/pose_estimator.py:98-125 solves one frame at
 *   a time).  rpe_pose_workspace_bytes covers every choice. */
typedef struct rpe_solve_opts {
    int struct_size;             /* sizeof(rpe_solve_opts) */
    int history_size;            /* 1..100 (torch.optim.LBFGS default 100) */
    double tolerance_grad;       /* 1e-7 */
    double tolerance_change;     /* 1e-9 */
    int partition_rows;          /* 0 = n */
    int reserved;                /* flags; 0 = defaults.  RPE_SOLVE_LAUNCH_PER_EVALUATION (bit 0): one launch per evaluation even where the whole
                                  * solve would run as one persistent launch (identical results; for A/B measurements) */
} rpe_solve_opts;
#define RPE_SOLVE_LAUNCH_PER_EVALUATION 1
int rpe_pose_solve_ex(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                      const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                      int n, int h, int w, int mode, int iters, const rpe_solve_opts *opts, double *T_out, float *vec7,
                      float *log6, int32_t *info, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Backward of the declarative pose layer (training): replaces DeclarativeNodeLie.gradient /
 * _get_objective_derivatives (core/optimization/declerative_node_lie.py:13-82,106-126; called from
 * DeclarativeFunctionLie.backward :249-267), which differentiates the objective twice with autograd, by the closed
 * forms of fY, fYY and fXY^T u (csrc/pose_backward.hip).  Inputs as rpe_pose_reduce; T (n,7) f64 = the layer's output
 * pose (the reference uses its float32 vec7).
 *   rpe_pose_backward_moments: out (n,48) f64 = [g2u(6) | g3u(6) | H(6x6 row-major)]: tangent gradients of the
 *       reprojection / 3-D terms with unit loss weight (fY = lw[1] g2u + lw[0] g3u; d fY/d loss_weight) and
 *       H = (fYY + fYY^T)/2 as the reference forms it (:51).  workspace: rpe_pose_backward_workspace_bytes(n,h,w).
 *   rpe_pose_backward_grads:   given u (n,6) f64 = -H^-1 v, writes fXY^T u for flow (n,2,h,w), pcl1, pcl2 (n,3,h,w),
 *       w1, w2 (n,1,h,w), float32, NaN -> 0 (:76); any output pointer may be NULL.
 * The 6x6 solve, the optimality test |fY| <= eps (:43-47) and d/d loss_weight = (u.g3u, u.g2u) stay on the host side. */
size_t rpe_pose_backward_workspace_bytes(int n, int h, int w);
int rpe_pose_backward_moments(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                              const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                              const double *T, int n, int h, int w, double *out, void *workspace, void *stream);
int rpe_pose_backward_grads(const float *flow, const float *pcl1, const float *pcl2, const float *w1, const float *w2,
                            const uint8_t *mask1, const uint8_t *mask2, const float *K, const float *loss_weight,
                            const double *T, const double *u, int n, int h, int w, float *grad_flow, float *grad_pcl1,
                            float *grad_pcl2, float *grad_w1, float *grad_w2, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stereo depth, back-projection, flow warps and the 1/8 stacks of the weight heads -- replaces
 * core/pose/pose_net.py:73-79 (depth from disparity, validity, proj), :104-108 (remap_from_flow x3,
 * remap_from_flow_nearest; core/interpol/flow_utils.py:4-26) and :110-113 (bilinear x0.125 of the two
 * 8-channel stacks) in ONE pass.  h and w must be multiples of 8.
 *
 *   in : stereo_flow2 (n,2,h,w)  time_flow (n,2,h,w)  baseline (n)  K (n,3,3)  depth1 (n,1,h,w)
 *        image1l, image2l (n,3,h,w)  stereo_flow1 (n,2,h,w)  mask2 (n,1,h,w) u8
 *   out: depth2 (n,1,h,w)  mask2_valid (n,1,h,w) u8 = mask2 & (0 < depth <= 1)   [pose_net.py:75-77]
 *        pcl1 (n,3,h,w)  pcl2w (n,3,h,w) = remap_from_flow(proj(depth2), time_flow)
 *        mask2w (n,1,h,w) u8 = valid_mapping & remap_nearest(mask2_valid)           [pose_net.py:107-108]
 *        inp1 (n,8,h/8,w/8) = down8(cat(stereo_flow1, image1l, pcl1))
 *        inp2 (n,8,h/8,w/8) = down8(cat(warp(stereo_flow2), warp(image2l), pcl2w))
 *        pcl2 (n,3,h,w) un-warped cloud, may be NULL
 * --------------------------------------------------------------------------------------------------------- */
int rpe_depth_backproject_warp(const float *stereo_flow2, const float *time_flow, const float *baseline,
                               const float *K, const float *depth1, const float *image1l, const float *image2l,
                               const float *stereo_flow1, const uint8_t *mask2, int n, int h, int w,
                               float *depth2, uint8_t *mask2_valid, float *pcl1, float *pcl2w, uint8_t *mask2w,
                               float *inp1, float *inp2, float *pcl2, void *stream);

/* flow2depth of core/pose/pose_net.py:127-135 (first frame): depth (n,1,h,w), valid (n,1,h,w) u8. */
int rpe_flow2depth(const float *stereo_flow, const float *baseline, int n, int h, int w, float *depth,
                   uint8_t *valid, void *stream);

/* Integer sampling taps of the two warps for index-parity tests: x0,y0 = floor (bilinear), xn,yn =
 * round-half-even (nearest) of the un-normalised grid_sample position.  Each (n,h,w) int32.            */
int rpe_warp_taps(const float *flow, int n, int h, int w, int32_t *x0, int32_t *y0, int32_t *xn, int32_t *yn,
                  void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * RAFT correlation -- replaces CorrBlock (core/RAFT/core/corr.py of the aimi-lab/RAFT submodule: all-pairs
 * correlation / sqrt(C), 4-level 2x2 average-pool pyramid, radius-r bilinear window lookup), called from
 * RAFT.forward (call sites core/pose/pose_net.py:47,65,129).
 *
 * The pyramid is an opaque device buffer owned by the caller; its internal layout (f32; the maps of 8 x-neighbouring
 * queries interleaved and skewed) is private to build/lookup.
 * --------------------------------------------------------------------------------------------------------- */
size_t rpe_corr_pyramid_bytes(int b, int h8, int w8, int levels);
/* the same for a given feature_dtype of rpe_corr_build_ex: RPE_F32 / RPE_F16 need rpe_corr_pyramid_bytes; the RPE_F32X3 experiment keeps
 * both feature maps as three bf16 planes and needs more scratch -- a pyramid that rpe_corr_build_ex(..., RPE_F32X3) writes MUST have been
 * sized with this function and RPE_F32X3 (the library sees only the pointer; the Python binding checks the buffer length) */
size_t rpe_corr_pyramid_bytes_ex(int b, int h8, int w8, int levels, int feature_dtype);
/* fmap1, fmap2: (b,c,h8,w8) f32.  Computes corr[b,q1,q2] = <fmap1[:,q1], fmap2[:,q2]> / sqrt(c) and the
 * average-pooled levels. */
int rpe_corr_build(const float *fmap1, const float *fmap2, int b, int c, int h8, int w8, int levels,
                   void *pyramid, void *stream);
/* Same with the precision of the feature maps selectable: feature_dtype = RPE_F32 (as rpe_corr_build) or RPE_F16 =
 * "fp16 features" (BASELINE config 5; RAFT's mixed_precision encoders hand fp16 feature maps to corr.py, which calls
 * .float() on them): both maps are rounded to fp16 (round to nearest even), the products run on the 16-bit matrix cores
 * with f32 accumulation (exact products, f32 sums) and the pyramid stays f32.  c % 16 == 0.
 * RPE_F32X3 (experiment switch): f32 maps, each f32 product evaluated as six bf16 products of an exact three-way split of both factors,
 * f32 accumulation -- f32-equivalent results (csrc/corr.hip, k_corr_build_x3). */
int rpe_corr_build_ex(const float *fmap1, const float *fmap2, int b, int c, int h8, int w8, int levels, int feature_dtype,
                      void *pyramid, void *stream);
/* coords (b,2,h8,w8) f32 (channel 0 = x, 1 = y) -> out (b, levels*(2r+1)^2, h8, w8) f32, channel order
 * level-major then window index i*(2r+1)+j with x offset (i-r) and y offset (j-r) (upstream's transposed
 * window).  radius must be 4, levels <= 4. */
int rpe_corr_lookup(const void *pyramid, const float *coords, int b, int h8, int w8, int levels, int radius,
                    float *out, void *stream);
/* The lookup FUSED into the convolution that consumes it: out (b, 256, h8, w8) = act(convc1(lookup(coords))) without the 324-channel
 * tensor ever reaching memory -- BasicMotionEncoder.forward's first layer (upstream core/RAFT/core/update.py: cor = F.relu(self.convc1(corr)),
 * 1x1, 324 -> 256; the reference's call site is core/pose/pose_net.py:65).  A workgroup looks the four levels of 64 queries up into LDS
 * and contracts them on the f32 matrix cores; the products are added in rpe_conv1x1's order, so the result is BIT-IDENTICAL to
 * rpe_corr_lookup followed by rpe_conv1x1 (or rpe_conv_fused) with the same weights.  packed = rpe_corr_lookup_conv1x1_pack of the
 * (256, 324, 1, 1) weight (rpe_corr_lookup_conv1x1_packed_floats floats; 0 for any other shape); bias (256) or NULL; relu != 0: ReLU;
 * out / out2 (may be NULL): channel slices given by their batch strides (floats).  levels must be 4, radius 4, w8 % 8 == 0; anything else
 * -> RPE_E_UNSUPPORTED (the caller runs the two entry points).  Launches of at most one 64-query tile per compute unit run one workgroup per
 * tile (35 us against 45 for the two kernels at 2 pairs of 640x512); larger ones run persistent workgroups that look tile n + 1 up under the
 * matrix phase of tile n (315 us against 307 at 32 pairs: the f32 matrix pipe bounds both, so the two entry points remain the route for
 * large batches; measured in NOTES.md, round 6).  rpe_corr_lookup stays the stand-alone entry point (and the kernel the HBM roofline of
 * bench.py is stated on). */
size_t rpe_corr_lookup_conv1x1_packed_floats(int cout, int cin);
int rpe_corr_lookup_conv1x1_pack(const float *weight, int cout, int cin, float *packed, void *stream);
int rpe_corr_lookup_conv1x1(const void *pyramid, const float *coords, int b, int h8, int w8, int levels, int radius, const float *packed,
                            const float *bias, int relu, float *out, long long out_batch_stride, float *out2, long long out2_batch_stride,
                            void *stream);
/* Integer taps of the lookup for index-parity tests: x0,y0 (b,levels,2r+1,h8*w8) int32 = floor of the
 * un-normalised grid_sample position of window tap i on each axis (x0[..,i,q] pairs with x offset i-r,
 * y0[..,j,q] with y offset j-r) -- the very indices rpe_corr_lookup reads.  -1000000 marks a non-finite tap. */
int rpe_corr_lookup_taps(const float *coords, int b, int h8, int w8, int levels, int32_t *x0, int32_t *y0,
                         void *stream);
/* Diagnostic for the roofline claim of rpe_corr_lookup (bench.py): for these coordinates, per (batch item, level, group of 8
 * x-neighbouring queries) the number of staging rounds the lookup takes (1 = the group's eight windows fit one 12 x 16 box:
 * smooth flow) and the 128-B pyramid lines its loader requests.  rounds, lines: (b, levels, h8 * ceil(w8/8)) int32.  Same
 * tap / round arithmetic as the lookup kernel (shared device code), no pyramid access. */
int rpe_corr_lookup_rounds(const float *coords, int b, int h8, int w8, int levels, int32_t *rounds, int32_t *lines,
                           void *stream);
/* Copy one level of the pyramid out as a dense (b*h8*w8, h8>>l, w8>>l) f32 tensor (tests only). */
int rpe_corr_export_level(const void *pyramid, int b, int h8, int w8, int levels, int level, float *dense,
                          void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * RAFT update block, element-wise halves of SepConvGRU (core/RAFT/core/update.py) fused around the
 * convolutions, and the convex 8x up-sampling of RAFT.upsample_flow (core/RAFT/core/raft.py).
 * --------------------------------------------------------------------------------------------------------- */
/* zr_pre (b,2c,hw): pre-activations (WITHOUT bias; zr_bias (2c) and the tensor zr_add (b,2c,hw) are added here, either
 * may be NULL) of the z and r convolutions stacked on the channel axis.  zr_add carries the loop-invariant part of
 * the convolution: the context features `inp` are the same in all GRU iterations, so conv(inp-channels) is computed
 * once per RAFT pass and added here instead of being recomputed 12 times.  h is read from
 * channels [0,c) of a (b,h_channels,hw) buffer.  Writes z = sigmoid(zr_pre[:, :c]) to z_out (b,c,hw) and
 * r*h = sigmoid(zr_pre[:, c:]) * h into channels [0,c) of rh_out (b,rh_channels,hw), so rh_out can be the
 * (r*h | x) buffer the q convolution reads. */
int rpe_gru_gates_zr(const float *zr_pre, const float *zr_bias, const float *zr_add, const float *h, int h_channels,
                     int b, int c, int hw, float *z_out, float *rh_out, int rh_channels, void *stream);
/* h_out = (1 - z) * h + z * tanh(q_pre + q_add + q_bias[c]) (q_bias (c), q_add (b,c,hw) may be NULL); h_out may alias h.  h / h_out are channels [0,c) of buffers with
 * h_channels / hout_channels channels per batch row, so the new state lands straight in the (h|x) buffer. */
int rpe_gru_gates_h(const float *z, const float *q_pre, const float *q_bias, const float *q_add, const float *h,
                    int h_channels, int b, int c, int hw, float *h_out, int hout_channels, void *stream);
/* y = act(x + bias[c]) for x (b,c,hw): the bias add, ReLU (relu != 0) and the torch.cat / copy_ that follow a
 * convolution of the update block (core/RAFT/core/update.py BasicMotionEncoder / FlowHead), in one pass.  The result
 * goes to channels [out1_offset, out1_offset+c) of out1 (b,out1_channels,hw) and, if out2 != NULL, also to out2.
 * bias may be NULL; out1 may alias x when out1_channels == c. */
int rpe_bias_act(const float *x, const float *bias, int b, int c, int hw, int relu, float *out1, int out1_channels,
                 int out1_offset, float *out2, int out2_channels, int out2_offset, void *stream);
/* Encoder epilogues (core/RAFT/core/extractor.py ResidualBlock.forward / BasicEncoder.forward): for x (b,c,hw)
 *   y = norm(x + bias[c]);  if (relu) y = max(y,0);  if (residual != NULL) y = max(residual + y, 0)
 * rpe_instnorm_act: per-(b,c) instance norm, biased variance, eps inside the root (torch.nn.InstanceNorm2d, fnet).
 * rpe_affine_act:   y = x*scale[c] + shift[c], i.e. an eval-mode BatchNorm2d folded with the conv bias (cnet).
 * out may alias x.  bias / residual may be NULL. */
int rpe_instnorm_act(const float *x, const float *bias, int b, int c, int hw, float eps, int relu,
                     const float *residual, float *out, void *stream);
int rpe_affine_act(const float *x, const float *scale, const float *shift, int b, int c, int hw, int relu,
                   const float *residual, float *out, void *stream);
/* dst[:, 0:c] = src[:, 0:c] for channel slices of NCHW buffers (pointer to the slice's first plane in batch item 0 + batch stride
 * in floats): the slice assignments / clones around RAFT's update loop (core/RAFT/core/raft.py: net = tanh(net), coords1 = coords0
 * .clone(), the returned hidden state) without a library copy kernel. */
int rpe_copy_planes(const float *src, long long src_batch_stride, float *dst, long long dst_batch_stride, int b, int c, int hw,
                    void *stream);
/* FlowHead.conv2 (core/RAFT/core/update.py) + the coordinate update of RAFT.forward (core/RAFT/core/raft.py):
 * out (b,2,h,w) = conv3x3(x (b,c,h,w), weight (2,c,3,3), padding 1) + bias (2) [+ add (b,2,h,w), may be NULL = plain
 * convolution; pass coords1 to get coords1 + delta_flow].  out may alias add.  Two output channels make this a
 * streaming problem, not a GEMM. */
int rpe_conv3x3_to2(const float *x, const float *weight, const float *bias, int b, int c, int h, int w,
                    const float *add, float *out, void *stream);
/* The same with the flow bookkeeping of RAFT.forward's loop (core/RAFT/core/raft.py: coords1 = coords1 + delta_flow;
 * flow = coords1 - coords0 with coords0 the integer pixel grid; the motion encoder concatenates flow behind its output) fused in:
 * coords_out = conv + bias + coords, and flow = coords_out - grid is written to flow_out (b,2,h,w) and to the two-plane channel
 * slices dst1 / dst2 (pointer to the first plane in batch item 0 + batch stride in floats; any of the three may be NULL). */
int rpe_conv3x3_to2_flow(const float *x, const float *weight, const float *bias, int b, int c, int h, int w,
                         const float *coords, float *coords_out, float *flow_out, float *dst1, long long dst1_batch_stride,
                         float *dst2, long long dst2_batch_stride, void *stream);
/* flow (b,2,h8,w8), mask (b,576,h8,w8) raw logits already scaled by .25 -> out (b,2,8*h8,8*w8). */
int rpe_upsample_convex(const float *flow, const float *mask, int b, int h8, int w8, float *out, void *stream);

/* ---- update-block convolutions (core/RAFT/core/update.py: BasicMotionEncoder convc1/convc2/convf2/conv, SepConvGRU
 * convz|convr/convq of both halves, FlowHead.conv1) as implicit GEMMs on the f32 matrix cores with their epilogues
 * fused.  Stride 1 (stride 2: see the descriptor), zero padding k/2 ("same"), NCHW, kernel height odd, kernel width 1, 3 or 5, w % 4 == 0
 * (otherwise RPE_E_UNSUPPORTED: the caller keeps the library convolution + rpe_bias_act / rpe_gru_gates_*).
 * Every tensor argument is a pointer to channel 0 of a channel slice plus the batch stride (in floats) of the
 * buffer it lives in, so inputs and outputs can be slices of the concatenated (h | motion | flow) buffers.
 *   v = conv(x)[co][p] * scale[co] + add[co][p] + bias[co]           (scale, add, bias: each may be NULL)
 *   RPE_CONV_LINEAR : y = v;  RPE_CONV_RELU : y = max(v, 0);  then, if residual != NULL, y = max(residual + y, 0)
 *   RPE_CONV_TANH   : y = tanh(v)   (rpe_conv_fused, plain epilogue only: no scale / residual / stats / stride 2)
 *                     (the encoder's ResidualBlock tail, core/RAFT/core/extractor.py); out = y (and out2 = y when
 *                     out2 != NULL).  If stats != NULL the kernel also writes per-tile moments of v,
 *                     stats[b][cout][rpe_conv_stats_tiles(cout,h,w,stride)][3] = (count, mean, sum of squared deviations
 *                     from that mean; taken about a pivot so that |mean| >> std loses no digits), for rpe_instnorm_apply
 *                     (instance norm in one further read + write pass instead of three).
 *   RPE_CONV_GATE_ZR: cout = 2*gate_channels; co <  gate_channels: out[co]  = sigmoid(v)                 (z)
 *                                             co >= gate_channels: out2[co-gate_channels] = sigmoid(v) * hidden[co-gate_channels]  (r*h)
 *   RPE_CONV_GATE_H : out[co] = (1 - zgate[co]) * hidden[co] + zgate[co] * tanh(v);  out may alias hidden.        */
#define RPE_CONV_LINEAR 0
#define RPE_CONV_RELU 1
#define RPE_CONV_GATE_ZR 2
#define RPE_CONV_GATE_H 3
#define RPE_CONV_TANH 4     /* y = tanh(v): the hidden-state half of the context encoder's output (core/RAFT/core/raft.py: net = tanh(net)) */
typedef struct rpe_conv_desc {
    const float *x;      long long x_batch_stride;      /* input slice (b, cin, h, w)                              */
    const float *packed;                                 /* weights from rpe_conv_pack                              */
    const float *bias;                                   /* (cout) or NULL                                          */
    const float *add;    long long add_batch_stride;    /* (b, cout, h, w) pre-activation addend or NULL           */
    float *out;          long long out_batch_stride;
    float *out2;         long long out2_batch_stride;   /* NULL unless described above                             */
    const float *hidden; long long hidden_batch_stride; /* gates only                                              */
    const float *zgate;  long long zgate_batch_stride;  /* RPE_CONV_GATE_H only                                    */
    const float *scale;                                  /* (cout) or NULL (folded batch norm: scale, bias = shift) */
    const float *residual; long long residual_batch_stride; /* LINEAR / RELU only                                   */
    float *stats;                                        /* LINEAR / RELU only                                      */
    const float *pre_norm;                               /* (b, cin, 2) = (mean, 1/std) from rpe_instnorm_finalize or NULL: x is the RAW output
                                                          * of the previous convolution and is normalised + ReLU'd while it is staged (3x3, stride 1) */
    int b, cin, cout, h, w, kh, kw, mode, gate_channels;  /* h, w: INPUT map                                       */
    int stride;                                          /* 0 or 1: stride 1; 2: the encoders' down-sampling convolutions (3x3 pad 1 or 1x1 pad 0,
                                                          * even h and w, LINEAR / RELU only); the output map is (h/2, w/2)                        */
    int stats_tiles;                                     /* records per (b, cout) plane of `stats`: 0 or rpe_conv_stats_tiles(cout,h,w,stride)
                                                          * (checked when non-zero; it selects nothing) */
} rpe_conv_desc;
/* number of floats of the packed form of a (cout, cin, kh, kw) weight tensor (0 on bad arguments) */
size_t rpe_conv_packed_floats(int cout, int cin, int kh, int kw);
/* weight (cout, cin, kh, kw) contiguous -> packed (tap-major 16-channel steps, output channels padded to 128) */
int rpe_conv_pack(const float *weight, int cout, int cin, int kh, int kw, float *packed, void *stream);
int rpe_conv_fused(const rpe_conv_desc *desc, void *stream);
/* The same operation for 3x3 stride-1 convolutions with LINEAR / RELU epilogues (even h and w, cin % 4 == 0) as Winograd
 * F(2x2,3x3) on the f32 matrix cores: 2.25x fewer matrix FLOPs than the direct form (csrc/conv_wino.hip; the update block's
 * convc2, convf2, conv and FlowHead.conv1, core/RAFT/core/update.py, and the encoders' residual blocks,
 * core/RAFT/core/extractor.py).  Supported descriptor fields: bias, scale, out, out2, residual, stats, pre_norm (cin <= 128
 * with the last four: the encoders' widths); stats then holds T = rpe_conv_wino_stats_tiles(h, w) records per plane laid out (b, T, cout, 3) -- pass
 * tiles = -T to rpe_instnorm_apply / rpe_instnorm_finalize (negative = tile-major layout).  desc->packed must come from
 * rpe_conv_wino_pack (rpe_conv_wino_packed_floats floats; 0 = unsupported shape).  add / gates -> RPE_E_UNSUPPORTED
 * (the caller uses rpe_conv_fused). */
size_t rpe_conv_wino_packed_floats(int cout, int cin);
int rpe_conv_wino_stats_tiles(int h, int w);
int rpe_conv_wino_pack(const float *weight, int cout, int cin, float *packed, void *stream);
int rpe_conv_wino(const rpe_conv_desc *desc, void *stream);
/* LABELLED VARIANT of rpe_conv_wino (never in a headline number; bench.py --conv-bf16x3): the same convolution, descriptor fields,
 * epilogues and moment records, with every f32 product of the Winograd domain evaluated as six bf16 products of an exact three-way
 * split (x = hi + mid + lo) on the 16-bit matrix cores, f32 accumulation: f32-equivalent error (tests/test_gpu_conv_x3.py holds it to
 * 1.25x the f32 kernels'), 3/8 of the matrix time (csrc/conv_wino_x3.hip).  Needs cin % 16 == 0, w % 4 == 0, even h, 16-byte
 * aligned tensors; anything else -> RPE_E_UNSUPPORTED (the caller uses rpe_conv_wino).  desc->packed must come from
 * rpe_conv_wino_x3_pack (rpe_conv_wino_x3_packed_bytes bytes, 0 = unsupported shape; 16-byte aligned).  Same reference layers as
 * rpe_conv_wino (core/RAFT/core/update.py, extractor.py; call sites core/pose/pose_net.py:47,65,129). */
size_t rpe_conv_wino_x3_packed_bytes(int cout, int cin);
int rpe_conv_wino_x3_pack(const float *weight, int cout, int cin, void *packed, void *stream);
int rpe_conv_wino_x3(const rpe_conv_desc *desc, void *stream);
/* The same operation for 1x5 / 5x1 stride-1 convolutions (w % 4 == 0, cin % 4 == 0; tensors 16-byte aligned) with every
 * epilogue mode of rpe_conv_fused incl. the GRU gates (add, hidden, zgate, out2, gate_channels), as Winograd F(4,5) along the
 * filter axis on the f32 matrix cores: 8 products per 4 outputs instead of 20 (csrc/conv_wino1d.hip; SepConvGRU's convz|convr
 * and convq, core/RAFT/core/update.py).  desc->packed must come from rpe_conv_wino1d_pack (weight (cout, cin, 5) = the
 * contiguous (cout, cin, 1, 5) or (cout, cin, 5, 1) tensor; rpe_conv_wino1d_packed_floats floats, 0 = unsupported shape);
 * kh, kw select the axis.  scale / residual / stats / pre_norm -> RPE_E_UNSUPPORTED. */
size_t rpe_conv_wino1d_packed_floats(int cout, int cin);
int rpe_conv_wino1d_pack(const float *weight, int cout, int cin, float *packed, void *stream);
int rpe_conv_wino1d(const rpe_conv_desc *desc, void *stream);
/* LABELLED VARIANT of rpe_conv_wino1d (never in a headline number; bench.py --conv-bf16x3): the same convolution, descriptor fields
 * and epilogue modes incl. the GRU gates, with every f32 product of the Winograd domain evaluated as six bf16 products of an exact
 * three-way split on the 16-bit matrix cores, f32 accumulation (csrc/conv_wino1d_x3.hip; error held to the f32 kernels' by
 * tests/test_gpu_conv_x3.py).  Needs cin % 16 == 0, w % 4 == 0, 16-byte aligned tensors; anything else -> RPE_E_UNSUPPORTED (the
 * caller uses rpe_conv_wino1d).  desc->packed must come from rpe_conv_wino1d_x3_pack (rpe_conv_wino1d_x3_packed_bytes bytes, 0 =
 * unsupported shape; 16-byte aligned).  Same reference layers as rpe_conv_wino1d (SepConvGRU, core/RAFT/core/update.py; call sites
 * core/pose/pose_net.py:47,65,129). */
size_t rpe_conv_wino1d_x3_packed_bytes(int cout, int cin);
int rpe_conv_wino1d_x3_pack(const float *weight, int cout, int cin, void *packed, void *stream);
int rpe_conv_wino1d_x3(const rpe_conv_desc *desc, void *stream);
/* 1x1 stride-1 convolutions as plain GEMMs with LDS-DMA operand rings (csrc/conv1x1.hip; BasicMotionEncoder.convc1 behind the
 * correlation lookup, the mask head's and the encoders' 1x1 output layers, core/RAFT/core/update.py / extractor.py).  Same
 * descriptor; supported fields: bias, out, out2, mode LINEAR / RELU / TANH; h * w % 4 == 0, 16-byte aligned input slice.  desc->packed
 * must come from rpe_conv1x1_pack (rpe_conv1x1_packed_floats floats).  Anything else -> RPE_E_UNSUPPORTED (use rpe_conv_fused). */
size_t rpe_conv1x1_packed_floats(int cout, int cin);
int rpe_conv1x1_pack(const float *weight, int cout, int cin, float *packed, void *stream);
int rpe_conv1x1(const rpe_conv_desc *desc, void *stream);
/* LABELLED VARIANT of rpe_conv1x1 (never in a headline number; bench.py --conv-bf16x3): the same GEMM with every f32 product as six bf16
 * products of a three-way split on the 16-bit matrix cores, f32 accumulation (csrc/conv1x1_x3.hip; BasicMotionEncoder.convc1,
 * core/RAFT/core/update.py; call sites core/pose/pose_net.py:47,65,129).  Same descriptor; supported: bias, out, out2, mode LINEAR / RELU;
 * h * w % 4 == 0, 16-byte aligned input and output slices; anything else -> RPE_E_UNSUPPORTED (use rpe_conv1x1).  desc->packed must come from
 * rpe_conv1x1_x3_pack (rpe_conv1x1_x3_packed_bytes bytes; 16-byte aligned).
 * SPECIAL VALUES (all three bf16x3 entry points; the f32 kernels do not share this): the split x = hi + mid + lo turns an infinite input into
 * hi = +-Inf, mid = Inf - Inf = NaN, so +-Inf in an activation or weight gives NaN in every output it reaches, where the f32 kernels give
 * +-Inf (or NaN only for Inf - Inf / 0 * Inf).  NaN inputs give NaN in both.  rpe_conv1x1_x3 with cin % 16 != 0 additionally reads channel
 * cin - 1 a second time against zero weights in its padded last K step: a NaN there is harmless (it is NaN anyway), an Inf falls under the
 * rule above.  Finite inputs, including denormals and values up to FLT_MAX / 4, are unaffected (tests/test_gpu_conv_x3.py). */
size_t rpe_conv1x1_x3_packed_bytes(int cout, int cin);
int rpe_conv1x1_x3_pack(const float *weight, int cout, int cin, void *packed, void *stream);
int rpe_conv1x1_x3(const rpe_conv_desc *desc, void *stream);
/* Generic convolution for every shape the tuned kernels above refuse (odd maps, rows that are not whole 16-byte quads): replaces
 * torch.nn.functional.conv2d(x, weight, bias, stride, padding) as the host code's fallback route, so that every convolution of the
 * reference's RAFT (core/RAFT/core/extractor.py, update.py) runs in this library for any img_size (configuration/infer_f2f.yaml:13).
 * x (b, cin, h, w) and out (b, cout, ho, wo) are channel slices given by their batch strides (floats); weight (cout, cin, kh, kw)
 * contiguous, bias (cout) or NULL; kh, kw <= 7; stride 1 or 2; zero padding (pad_h, pad_w); ho = (h + 2 pad_h - kh) / stride + 1;
 * relu != 0: out = max(., 0).  f32 matrix cores with element-wise operand gathers: robust, 2-3x slower than the tuned kernels. */
int rpe_conv_direct(const float *x, long long x_batch_stride, const float *weight, const float *bias, int b, int cin, int cout,
                    int h, int w, int kh, int kw, int stride, int pad_h, int pad_w, int relu, float *out,
                    long long out_batch_stride, void *stream);
/* number of moment records per (b, channel) plane rpe_conv_fused leaves for this shape (h, w: input): stride 1 one per pixel
 * tile (the launcher and this function share one tile-width rule); stride 2 one per 32 output pixels -- in both of its tile
 * classes (128 x 128, or 64 x 64 for launches too small to fill the chip), bit-identical between them, so an image's statistics
 * do not depend on how many images share the launch */
int rpe_conv_stats_tiles(int cout, int h, int w, int stride);
/* round-3 name, kept: the count no longer depends on the batch (returns rpe_conv_stats_tiles; 0 for b <= 0) */
int rpe_conv_stats_tiles_batch(int cout, int h, int w, int stride, int b);
/* Instance norm (torch.nn.InstanceNorm2d, affine=False; fnet of core/RAFT/core/extractor.py) of x (b,c,hw) given the
 * per-tile (count, mean, M2) records rpe_conv_fused / rpe_stem_conv left in `partials` (b,c,tiles,3) -- or rpe_conv_wino in
 * (b,|tiles|,c,3) when tiles < 0 --, merged in f64 with the parallel-variance formula:
 *   y = (x - mean) / sqrt(var + eps); if (relu) y = max(y,0); if (residual) y = max(residual + y, 0).  out may alias x. */
int rpe_instnorm_apply(const float *x, const float *partials, int tiles, int b, int c, int hw, float eps, int relu,
                       const float *residual, float *out, void *stream);
/* The same with a residual that is itself the RAW output of an instance-normalised convolution: residual_mean_inv (b,c,2) =
 * (mean, 1/sqrt(var + eps)) from rpe_instnorm_finalize, and the residual term becomes relu((residual - mean) * inv) -- the first
 * residual block of fnet then reads the stem's raw output twice (as input through `pre_norm`, as shortcut through this) and the
 * stem's normalised output is never written (BasicEncoder.forward: relu1(norm1(conv1 x)) -> layer1, core/RAFT/core/extractor.py).
 * relu: bit 0 = ReLU on y (as above); bit 1 = the raw residual is normalised WITHOUT a ReLU -- the shortcut of a stride-2 ResidualBlock,
 * norm3(conv1x1 x) (extractor.py), whose normalised output then never exists as a tensor either.
 * tiles == 0: `partials` is not records but the (b,c,2) (mean, 1/sqrt(var + eps)) pairs rpe_instnorm_finalize made of them (eps is
 * then unused): the pass is a pure stream, several workgroups per plane -- the faster form behind large maps. */
int rpe_instnorm_apply_ex(const float *x, const float *partials, int tiles, int b, int c, int hw, float eps, int relu,
                          const float *residual, const float *residual_mean_inv, float *out, void *stream);
/* mean_inv (b,c,2) = (mean, 1/sqrt(var + eps)) of each plane from the same partial sums: the `pre_norm` input of the
 * next rpe_conv_fused, which then normalises its input on the fly (no separate pass for norm1 + ReLU of a ResidualBlock). */
int rpe_instnorm_finalize(const float *partials, int tiles, int b, int c, int hw, float eps, float *mean_inv, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Prepared launch lists -- replace the HOST loop around the kernels above: RAFT.forward's `for itr in range(iters)` over lookup ->
 * motion encoder -> SepConvGRU -> flow head (core/RAFT/core/raft.py; call sites core/pose/pose_net.py:47,65,129) and the encoders'
 * layer-by-layer walk (core/RAFT/core/extractor.py), which the reference runs as hundreds of Python-dispatched launches per frame
 * (scripts/infer_trajectory.py:57,71-77 tracks one frame at a time: ~330 launches of 10-50 us each, i.e. the host decides the frame rate).
 * A list is an array of rpe_op in caller-owned host memory: each op names an entry point of this header (`kind`), the argument block that
 * entry point would be called with (`args`: the rpe_conv_desc of the convolutions, one of the rpe_*_args structs below for the others --
 * the same values in the same order as the function's parameters) and the stream it goes to (`stream` = index into the `streams` array of
 * the call).  rpe_run_ops walks the list once, calling the SAME entry points -- same checks, same kernels, same bits as calling them one
 * by one -- and stops at the first op that fails (its index goes to *failed_op, its status is returned).  The list and everything it
 * points to must stay alive and unchanged for the duration of the call only (launches are asynchronous; the argument blocks are consumed
 * at enqueue time); the caller may patch pointers inside the argument blocks between calls (new input / output buffers).
 * Fork / join between the streams of a list: RPE_OP_EVENT_RECORD records, RPE_OP_STREAM_WAIT makes streams[op.stream] wait for, the
 * hipEvent_t whose handle is stored at `args` (args = address of a void* cell holding the handle; a NULL handle makes the op a no-op, so a
 * caller can keep timing events in a list and arm them only when it measures).  No state is kept between calls. */
#define RPE_OP_CONV_FUSED 1        /* args: const rpe_conv_desc *  -> rpe_conv_fused       */
#define RPE_OP_CONV_WINO 2         /*       const rpe_conv_desc *  -> rpe_conv_wino        */
#define RPE_OP_CONV_WINO1D 3       /*       const rpe_conv_desc *  -> rpe_conv_wino1d      */
#define RPE_OP_CONV1X1 4           /*       const rpe_conv_desc *  -> rpe_conv1x1          */
#define RPE_OP_CONV_WINO_X3 5      /*       const rpe_conv_desc *  -> rpe_conv_wino_x3     */
#define RPE_OP_CONV_WINO1D_X3 6    /*       const rpe_conv_desc *  -> rpe_conv_wino1d_x3   */
#define RPE_OP_CONV1X1_X3 7        /*       const rpe_conv_desc *  -> rpe_conv1x1_x3       */
#define RPE_OP_CORR_LOOKUP 8       /*       const rpe_corr_lookup_args *                   */
#define RPE_OP_STEM_CONV 9         /*       const rpe_stem_conv_args *                     */
#define RPE_OP_FLOW_UPDATE 10      /*       const rpe_flow_update_args * -> rpe_conv3x3_to2_flow */
#define RPE_OP_COPY_PLANES 11      /*       const rpe_copy_planes_args *                   */
#define RPE_OP_INSTNORM_FINALIZE 12 /*      const rpe_instnorm_finalize_args *             */
#define RPE_OP_INSTNORM_APPLY 13   /*       const rpe_instnorm_apply_args * -> rpe_instnorm_apply_ex */
#define RPE_OP_UPSAMPLE_CONVEX 14  /*       const rpe_upsample_convex_args *               */
#define RPE_OP_CORR_BUILD 15       /*       const rpe_corr_build_args * -> rpe_corr_build_ex */
#define RPE_OP_LOOKUP_CONV1X1 16    /*       const rpe_lookup_conv1x1_args * -> rpe_corr_lookup_conv1x1 */
#define RPE_OP_EVENT_RECORD 32     /*       void *const * (address of a hipEvent_t handle; NULL handle = no-op) */
#define RPE_OP_STREAM_WAIT 33      /*       void *const * (the same)                       */
typedef struct rpe_op {
    int kind;                      /* RPE_OP_*                                             */
    int stream;                    /* index into rpe_run_ops' streams[]                    */
    const void *args;
} rpe_op;
typedef struct rpe_corr_lookup_args { const void *pyramid; const float *coords; int b, h8, w8, levels, radius; float *out; } rpe_corr_lookup_args;
typedef struct rpe_lookup_conv1x1_args {
    const void *pyramid; const float *coords; int b, h8, w8, levels, radius; const float *packed, *bias; int relu; float *out;
    long long out_batch_stride; float *out2; long long out2_batch_stride;
} rpe_lookup_conv1x1_args;
typedef struct rpe_corr_build_args { const float *fmap1, *fmap2; int b, c, h8, w8, levels, feature_dtype; void *pyramid; } rpe_corr_build_args;
typedef struct rpe_stem_conv_args {
    const float *image; int b, cin, h, w, stride; float div, mul, sub; const float *packed; int cout; const float *bias, *scale; int relu;
    float *out, *stats;
} rpe_stem_conv_args;
typedef struct rpe_flow_update_args {
    const float *x, *weight, *bias; int b, c, h, w; const float *coords; float *coords_out, *flow_out, *dst1; long long dst1_batch_stride;
    float *dst2; long long dst2_batch_stride;
} rpe_flow_update_args;
typedef struct rpe_copy_planes_args { const float *src; long long src_batch_stride; float *dst; long long dst_batch_stride; int b, c, hw; } rpe_copy_planes_args;
typedef struct rpe_instnorm_finalize_args { const float *partials; int tiles, b, c, hw; float eps; float *mean_inv; } rpe_instnorm_finalize_args;
typedef struct rpe_instnorm_apply_args {
    const float *x, *partials; int tiles, b, c, hw; float eps; int relu; const float *residual, *residual_mean_inv; float *out;
} rpe_instnorm_apply_args;
typedef struct rpe_upsample_convex_args { const float *flow, *mask; int b, h8, w8; float *out; } rpe_upsample_convex_args;
/* streams: n_streams hipStream_t handles (NULL = the default stream).  failed_op may be NULL. */
int rpe_run_ops(const rpe_op *ops, int n_ops, void *const *streams, int n_streams, int *failed_op);

/* ---- The two weight heads of PoseNet (core/pose/pose_net.py:109-115: torch.cat of the 1/8 stacks with the GRU hidden
 * state and the context -> TinyUNet(264) / TinyUNet(272), core/unet/unet.py:7-82 -> bilinear resize to (H, W) -> Sigmoid)
 * as one chain of 15 launches for both heads and all frames (csrc/unet.hip), inference-mode batch norm folded.
 *   inp1, inp2 (b,8,h8,w8) from rpe_depth_backproject_warp; hidden, context: channel 0 of (b,128,h8,w8) slices with their
 *   batch strides (floats); params2d / params3d: rpe_unet_params_floats(264 / 272) floats in the layout documented in
 *   csrc/unet.hip (3x3 weights as [cin][9][cout], batch norm as per-channel scale/shift); out2d, out3d (b,1,H,W) in (0,1).
 *   The 1/8 grid must be at least 44x44 (valid convolutions), as in the reference; otherwise RPE_E_UNSUPPORTED. */
size_t rpe_unet_params_floats(int in_channels);
size_t rpe_unet_workspace_bytes(int b, int h8, int w8);
int rpe_unet_heads(const float *inp1, const float *inp2, const float *hidden, const float *context, long long hidden_batch_stride,
                   long long context_batch_stride, const float *params2d, const float *params3d, int b, int h8, int w8, int H, int W,
                   float *out2d, float *out3d, void *workspace, void *stream);

/* ---- 7x7 convolutions of few input channels as patch-staged implicit GEMMs (csrc/stem.hip):
 *   cin = 3, stride 2: the encoders' first layer (core/RAFT/core/extractor.py BasicEncoder.conv1/norm1/relu1) on the RAW
 *                      0..255 image with RAFT.forward's normalisation image = 2 * (image / 255) - 1 (core/RAFT/core/raft.py)
 *                      applied while the input patch is staged (zero padding of the NORMALISED image, as the reference);
 *   cin = 2, stride 1: the motion encoder's convf1 on the flow (core/RAFT/core/update.py), div = mul = 1, sub = 0.
 * xn = mul * (x / div) - sub;  v = conv7x7(xn; pad 3) * scale[co] + bias[co] (scale NULL = 1: folded batch norm for cnet);
 * optional ReLU; stats (b,cout,rpe_stem_tiles(h,w,stride),3) receives per-tile (count, mean, M2) of v for
 * rpe_instnorm_apply (fnet).  cout % 64 == 0; stride 2 needs even h, w.  packed = rpe_stem_pack of the (cout,cin,7,7)
 * weight (rpe_stem_packed_floats floats). */
int rpe_stem_tiles(int h, int w, int stride);
size_t rpe_stem_packed_floats(int cout, int cin);
int rpe_stem_pack(const float *weight, int cout, int cin, float *packed, void *stream);
int rpe_stem_conv(const float *image, int b, int cin, int h, int w, int stride, float div, float mul, float sub,
                  const float *packed, int cout, const float *bias, const float *scale, int relu, float *out,
                  float *stats, void *stream);

/* ---- input side: what the reference's datasets do on the CPU before a frame reaches
 * PoseEstimator (dataset/stereo_dataset.py:12-16,35-40, dataset/video_dataset.py:59-63, dataset/transforms.py:20-39).
 * mask_specularities: out (h,w) u8 = erode_11x11((r+g+b < sum_threshold) & mask) on the decoded uint8 (h,w,3) image;
 * mask may be NULL; sum_threshold = ceil(3*255*spec_thr) (numpy compares the integer sum with a float); pixels outside
 * the image never erode (cv2.erode's default border value). */
int rpe_mask_specularities(const uint8_t *img_hwc, const uint8_t *mask, int h, int w, int sum_threshold, uint8_t *out,
                           void *stream);
/* ResizeStereo on one image: bilinear resize (torchvision 0.14 on tensors: align_corners=False, no antialias) of the
 * input to (resized_h, resized_w), then the centre crop [top, top+out_h) x [left, left+out_w); out (c,out_h,out_w) f32.
 * Input: float32 (c,h,w), or -- in_is_u8_hwc -- the decoded uint8 (h,w,c) image (fuses .permute(2,0,1).float()). */
int rpe_resize_crop(const void *in, int in_is_u8_hwc, int c, int h, int w, int resized_h, int resized_w, int top, int left,
                    int out_h, int out_w, float *out, void *stream);
/* the same with nearest sampling for the (h,w) u8 mask (InterpolationMode.NEAREST) */
int rpe_resize_crop_mask(const uint8_t *in, int h, int w, int resized_h, int resized_w, int top, int left, int out_h,
                         int out_w, uint8_t *out, void *stream);
/* Rectification of one image with precomputed maps: cv2.remap(src, mapx, mapy, INTER_NEAREST) with the default constant-0
 * border (dataset/preprocess/stereo_rectify.py:44-51, called per frame by StereoRectifier.__call__,
 * dataset/rectification.py:52-65): dst(ch,y,x) = src(ch, cvRound(mapy[y,x]), cvRound(mapx[y,x])), cvRound = round half to
 * even, saturated to int16; 0 outside the image.  src / dst planar (c,h,w) / (c,out_h,out_w), uint8 or float32
 * (src_is_u8); the maps are the CV_32FC1 pair initUndistortRectifyMap produces (preprocess.StereoRectifier builds them). */
int rpe_remap_nearest(const void *src, int src_is_u8, int c, int h, int w, const float *mapx, const float *mapy, int out_h,
                      int out_w, void *dst, void *stream);

/* Pseudo-rectification of the right image (dataset/rectification.py:55-58 mode='pseudo' -> dataset/preprocess/stereo_rectify.py:
 * 52-59 pseudo_rectify_2d): dst = cv2.warpAffine(src, [[1,0,tx],[0,1,ty]], (w,h)), i.e. INTER_LINEAR with source coordinates
 * rounded to 1/32 pixel (fixed point, AB_BITS = 10), OpenCV's 5-bit bilinear table (uint8: integer weights summing to 2^15,
 * (sum + 2^14) >> 15; float32: float weights), BORDER_CONSTANT 0.  tx = lkmat[0][2] - rkmat[0][2], ty = lkmat[1][2] - rkmat[1][2]
 * as float32 (the reference builds the matrix with .astype(np.float32)).  Planar (c,h,w) uint8 or float32.
 * RESTATED, UNPINNED: the arithmetic above is OpenCV 4.x's imgwarp.cpp as published, restated without cv2 in the build image; it is
 * tested bit for bit against a scalar restatement of the same description (oracle/rectify.py), NOT against cv2.warpAffine itself --
 * a 1-LSB difference to a particular OpenCV build (other interpolation tables, FMA contraction in remapBilinear<float>) would go
 * unnoticed until a golden from real cv2 (oracle/gen_golden.py on a machine that has it) pins both. */
int rpe_shift_bilinear(const void *src, int src_is_u8, int c, int h, int w, float tx, float ty, void *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RPE_H */
