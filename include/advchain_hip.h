/* advchain_hip.h -- C ABI of libadvchain_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ONE hot path of cherise215/advchain: the adversarial-augmentation
 * inner loop (ComposeAdversarialTransformSolver over AdvNoise / AdvBias / AdvMorph / AdvAffine).
 * The reference has no native layer of its own: its arithmetic is a set of PyTorch ATen call
 * sites.  Each entry point below replaces one (or a fused run) of those call sites; the
 * "replaces" line cites the reference file:line (paths relative to the upstream repo root).
 *
 * Contract (all entry points)
 *   - extern "C", plain pointers and sizes, no torch types.  Pointers are DEVICE pointers to
 *     contiguous fp32 (int32 for index tables) owned by the caller; nothing is allocated or
 *     freed inside; outputs / workspaces are caller-allocated.
 *   - layout: channels-first (N, C, S0, S1[, S2]); `ndim` = 2 or 3 spatial dims, `dims` has
 *     `ndim` entries in tensor order (slowest first).  Sampling grids are PLANAR (N, ndim, ...)
 *     with channel 0 = x <-> last spatial dim, 1 = y, 2 = z (F.grid_sample convention).  The sampler entry
 *     points (grid_sample, compose_self, expo_chain, affine_warp) take volumes of at least two voxels.
 *   - align_corners=True everywhere (the reference never uses False for samplers).
 *   - `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); kernels are
 *     enqueued asynchronously on it; the call never synchronises.
 *   - returns 0 on success, <0 on error (ADVCHAIN_ERR_*); advchain_last_error() gives the
 *     message (thread-local).  No exceptions cross the ABI.  Thread-safe for distinct
 *     streams/buffers.
 *   - backward entry points whose targets are scatter destinations (grad_in / grad_phi) require
 *     the caller to zero them first; they accumulate with hardware fp32 atomics.
 */
#ifndef ADVCHAIN_HIP_H_
#define ADVCHAIN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVCHAIN_INTERP_LINEAR 0  /* 'bilinear' (4-D) / trilinear (5-D) */
#define ADVCHAIN_INTERP_NEAREST 1
#define ADVCHAIN_INTERP_BICUBIC 2  /* 2D only; its own entry points (advchain_grid_sample_bicubic2d_*) */
#define ADVCHAIN_PAD_ZEROS 0
#define ADVCHAIN_PAD_BORDER 1
#define ADVCHAIN_PAD_REFLECTION 2

int advchain_version(void);
const char* advchain_last_error(void);

/* ---- deterministic mode (round 6) ---------------------------------------------------------
 * replaces: nothing the reference writes down -- its CPU path (the parity target) is deterministic by construction, and
 *           on a GPU it inherits torch.use_deterministic_algorithms(), under which grid_sampler_{2,3}d_backward
 *           (reached from adv_morph.py:546-557 through autograd) raises for want of a deterministic kernel.  Here the one
 *           formulation whose bits depend on arrival order -- the source-tiled window scatter (2D image warps above
 *           16 px, squarings above 32 px; 3D above 4 voxels), which flushes its LDS windows with float atomics -- gets a
 *           bit-reproducible twin: the tiles add 64-bit fixed point (2^40 / max|grad_out| of the batch entry) into an
 *           int64 image of grad_in inside the caller's workspace and one more pass converts it.  Every other backward
 *           formulation (gather forms, owner-computes scatters, affine tiles) is deterministic already.
 * PROCESS-WIDE switch, read when a backward entry is called and by advchain_scatter_workspace (which then returns the larger
 * size: allocate workspaces AFTER setting the mode; a workspace sized in the other mode must not be reused).  Not covered
 * (still float atomics): the overflow list of the LDS-tiled scatter (reflection padding, 3-channel image warps, calls without
 * a displacement bound), nearest-neighbour / size-changing backward, the bicubic backward, affine samples flagged as
 * degenerate, and the VALUE of the consistency loss / of the 3D step-count norm (partial sums arrive in any order; nothing
 * downstream of them but the number itself depends on the order).                                                       */
void advchain_set_deterministic(int on);
int advchain_get_deterministic(void);

/* ---- dense-field warp ------------------------------------------------------------------
 * replaces: F.grid_sample(data, grid.permute(..), mode, padding_mode, align_corners=True)
 *           advchain/augmentor/adv_morph.py:546-557 (AdvMorph.transform), and the final
 *           torch.clamp(dxy,-1,1) of adv_morph.py:304-305,490 when clamp_grid != 0 (the
 *           clamp and its sub-gradient mask are applied to the grid on load).
 * in (N,C,in_dims), grid (N,ndim,out_dims) planar, out (N,C,out_dims).                      */
/* clamp_grid: bit 0 = clamp the grid to [-1, 1] on load; bits 8..15 (optional) = the caller's estimate of the grid's
 * displacement |position - own voxel| in voxels, rounded up, 0 = unknown -- a performance hint that selects the forward
 * kernel (z-marching below one voxel, LDS tiles above); results do not depend on it.  advchain_compose_self_fwd takes the
 * same hint in bits 8..15 of final_mode. */
int advchain_grid_sample_fwd(const float* in, const float* grid, float* out, int64_t N, int64_t C, int ndim,
                             const int64_t* in_dims, const int64_t* out_dims, int interp, int padding,
                             int clamp_grid, void* stream);
/* The call above for two tensors through ONE grid: out = warp(in) (C channels) and ride_out = warp(ride_in) (one channel,
 * (N,1,in_dims) -> (N,1,out_dims)).
 * replaces: the pair of F.grid_sample calls the reference's solver makes with one deformation -- the data warp
 *           (adv_compose_solver.py:148-176 forward / 199-219 backward -> adv_morph.py:546-557) and the warp of the all-ones
 *           validity mask through the same transform (adv_compose_solver.py:262-268, 321-325).  2D: one launch (the taps of
 *           a sample are built once); 3D: the two launches of advchain_grid_sample_fwd.  Values are those of two separate
 *           calls, bit for bit.  flags bit 0: ride_out = (warp(ride_in) != 0) as 0 / 1 (the `masks[masks != 0] = 1` of
 *           adv_compose_solver.py:266-268 after the last warp of the round trip).                                      */
int advchain_grid_sample_fwd_ride(const float* in, const float* grid, float* out, const float* ride_in, float* ride_out,
                                  int64_t N, int64_t C, int ndim, const int64_t* in_dims, const int64_t* out_dims,
                                  int interp, int padding, int clamp_grid, int flags, void* stream);
/* replaces: autograd grid_sampler_{2,3}d_backward for the call above.
 * grad_in (N,C,in_dims) and grad_grid (N,ndim,out_dims); either may be NULL.  grad_grid is overwritten.
 * With `workspace` (int32[advchain_scatter_workspace(N,ndim,dims)]) grad_in is simply overwritten: the
 * LDS-tiled owner-computes scatter (64-bit fixed-point accumulation, deterministic) is used for linear
 * interpolation with in_dims == out_dims and C <= 4, and grad_in is zero-filled internally otherwise;
 * with workspace == NULL the global-atomic path is used and grad_in must be pre-zeroed by the caller.
 * halo > 0: an upper bound on the displacement |sampling position - own voxel| in voxels (see
 * advchain_max_displacement) -- a performance hint only: small bounds (1 in 3D, <= 4 in 2D; C in {1,4}) select the
 * gather-form adjoint (no atomics), larger ones size the tile halo; samples beyond the bound stay correct through the
 * overflow list.  Bounds above the gather form (2D: any; 3D: hints >= 2, i.e. one voxel and more) select the
 * source-tiled window scatter (float atomics between tiles: summation order, ~1e-7 relative, varies run to run).
 * 0 = default tiles.  halo < 0: |halo| is exact (guaranteed by the caller): see advchain_compose_self_bwd.  */
int64_t advchain_scatter_workspace(int64_t N, int ndim, const int64_t* dims); /* int32 elements (larger in deterministic mode) */
int advchain_grid_sample_bwd(const float* grad_out, const float* in, const float* grid, float* grad_in,
                             float* grad_grid, int32_t* workspace, int64_t N, int64_t C, int ndim,
                             const int64_t* in_dims, const int64_t* out_dims, int interp, int padding, int clamp_grid,
                             int halo, void* stream);

/* ---- scaling-and-squaring step ---------------------------------------------------------
 * replaces: applyComposition{2,3}D(phi, phi) = F.grid_sample(phi, phi^T, 'border',
 *           align_corners=True), adv_morph.py:179-202, called 8+ times from
 *           vectorFieldExponentiation{2,3}D adv_morph.py:132-135,165-168.
 * final_mode 1 additionally emits (sample - phi0) + identity, i.e. 'phi - grid_wh' (with the
 * in-place aliasing of adv_morph.py:111,143,176) plus '+ self.base_grid' of :474,483.
 * phi, out, phi0: (N, ndim, dims).
 * disp_out (may be NULL): ADVCHAIN_DISP_SLOTS (4096) floats, zero-initialised by the caller; the kernel
 * max-accumulates (atomically, spread over the slots) the displacement |position - own voxel| of `out` in voxels --
 * the max over the slots is the bound the backward of the NEXT squaring wants (advchain_compose_self_bwd: halo). */
#define ADVCHAIN_DISP_SLOTS 4096
int advchain_compose_self_fwd(const float* phi, float* out, const float* phi0, int64_t N, int ndim,
                              const int64_t* dims, int final_mode, float* disp_out, void* stream);
/* grad_phi receives both the value path (scatter) and the coordinate path; overwritten when a
 * `workspace` (as above) is given, otherwise it must be pre-zeroed (global-atomic path).
 * chain != 0: grad_out is the grad_phi of the previous call on the same workspace (the backward of
 * consecutive squarings), whose max|.| is already in the workspace -- saves one pass over grad_out.
 * halo > 0: an upper bound on |displacement| of phi in voxels -- a performance hint only: samples beyond it stay
 * correct through the overflow list (global atomics).  Small bounds (1 in 3D; 1..4 in 2D) select the gather-form
 * adjoint (no atomics, bit-reproducible); larger ones the source-tiled window scatter (float atomics between tiles; the
 * LDS-tiled fixed-point scatter remains for reflection padding, C = 3 image warps and ADVCHAIN_WINDOW3D_MIN_HALO);
 * 0 = default scatter tiles (halo 2 in 3D) / window scatter (2D).
 * halo < 0: |halo| is EXACT -- the caller guarantees no sample moves |halo| voxels or more on any axis (measured with
 * advchain_max_displacement / disp_out): the gather form then skips the overflow list and is a single launch; samples
 * violating the guarantee would be dropped.  Exact bounds beyond the gather form (3D: 2..4 voxels; 2D: 4, 8, 16 px, squarings also 32 px; both
 * entries) select the owner-computes scatters of scatter_march.hip: LDS 32-bit fixed-point accumulators scaled by the
 * max|grad_out| over the rows a workgroup visits, plain stores, no zero fill, bit-reproducible, relative error of the
 * accumulation <= 2^-23 / 2^-22 / 2^-21 (bounds of 2 / 3 / 4; 2^-18 for 5..8) of that local maximum per deposit; a NaN / inf in
 * grad_out turns the outputs of the workgroups that visit its row into NaN (it is not dropped).  (Launches that do not track max|result| -- strict gather,
 * window scatter, owner-computes scatters -- leave a marker in the workspace, and a chained scatter call after them
 * recomputes max|grad_out| on the device.) */
int advchain_compose_self_bwd(const float* grad_out, const float* phi, float* grad_phi, int32_t* workspace, int chain,
                              int halo, int64_t N, int ndim, const int64_t* dims, void* stream);
/* replaces: the loops of vectorFieldExponentiation{2,3}D (adv_morph.py:132-135,165-168) and their autograd, n squarings
 * in one call -- exactly the launches of n calls of advchain_compose_self_fwd / _bwd (same kernels, same results).
 * fwd: phi_1..phi_{n-1} -> fields[(n-1)][N][ndim][dims], the sampling positions (final_mode 1 of the last squaring)
 *      -> pos; disp_rows (may be NULL): (n+1) x ADVCHAIN_DISP_SLOTS zero-initialised floats, row m receives the
 *      displacement of phi_m (m >= 1; row 0 belongs to whoever made phi0), row n that of pos; hints (may be NULL): n
 *      displacement estimates in voxels (0 = unknown) for phi_0..phi_{n-1} (bits 0..7: as for advchain_grid_sample_fwd;
 *      bits 8..: the same estimate in 1/1024 voxel, 0 = unknown -- read by the 2D fusing rule).
 *      fuse_flag (may be NULL; TWO 32-bit words, ZERO before the call: a float flag and the arrival counter of the repeat
 *      launch's grid barrier): 2D only -- allows the leading squarings whose hinted input
 *      displacement is below one pixel, and the squarings behind them while the row halos of all fused levels (a level whose
 *      input moves less than h pixels needs h rows) sum to at most 5 (at most 5 levels, rows of 64k <= 512 pixels), to run as
 *      ONE launch over whole-row LDS windows (same arithmetic, bit-identical fields).  The kernel checks the premise on every window, level by level; a
 *      window that has to stop early records how many levels it could not do (*fuse_flag = the largest such count), and ONE
 *      repeat launch enqueued behind it (a persistent grid that returns at once while the flag is down) runs exactly those
 *      levels the ordinary way -- results never depend on the hints.  The caller may read the flag back (> 0: the hints were too optimistic) and must zero it
 *      before the next call.
 * bwd: grad_pos -> grad_phi0 through the n adjoint steps; halos[i] is the bound for the i-th step in BACKWARD order
 *      (squaring n-1 first); scratch: one field-sized buffer; workspace as for advchain_compose_self_bwd.             */
/* Query (host only, nothing is enqueued): how many leading squarings advchain_expo_chain_fwd would run as ONE fused launch for
 * this shape and these hints (0: none -- 3D, no hints, fewer than 256 windows, rows that are not a multiple of 64 pixels ...;
 * the pointer-alignment condition of the launch is not part of the answer).  Lets a caller / a test state which formulation
 * a chain takes; the reference has no counterpart (adv_morph.py:132-135 is one loop).                                  */
int advchain_expo_chain_fused_levels(int64_t N, int ndim, const int64_t* dims, int n, const int32_t* hints);
int advchain_expo_chain_fwd(const float* phi0, float* fields, float* pos, int64_t N, int ndim, const int64_t* dims, int n,
                            float* disp_rows, const int32_t* hints, float* fuse_flag, void* stream);
int advchain_expo_chain_bwd(const float* grad_pos, const float* phi0, const float* fields, float* grad_phi0, float* scratch,
                            int32_t* workspace, const int32_t* halos, int64_t N, int ndim, const int64_t* dims, int n,
                            void* stream);
/* Composite entries (round 5): ONE call enqueues a whole DemonsCompose direction for the paired field [v; -v] of a solver step
 * -- replaces: AdvMorph.DemonsCompose (adv_morph.py:454-491: Gaussian of the low-resolution velocity, F.interpolate, the
 * exponentiation of :116-146, composition with the identity grid, the final Gaussian) as the reference's forward() / backward()
 * call it (adv_morph.py:299-303,322-324), and its autograd.  2D, the 9-tap window, low-resolution planes of at most 4096
 * values.  The launches are exactly those of advchain_gauss_small_pair, advchain_tp_interp_fwd, advchain_expo_chain_fwd,
 * advchain_gauss_xy, advchain_slot_rows_max (forward) and advchain_gauss_xy, advchain_expo_chain_bwd,
 * advchain_band_reduce_rows_dense / _axis, advchain_gauss_small_pair (backward) in that order: same results, one trip out of
 * the host language per direction instead of five or six.  Every buffer is the caller's: s1 (2N, d, g...), phi0 / pos / q /
 * gpos / g / scratch (2N, d, S...), fields (n-1, 2N, d, S...), disp ((n+2) x ADVCHAIN_DISP_SLOTS, zero) + rows_max (n+2) or both
 * NULL, t1 (2N, d, S0, g1), gs1 (2N, d, g...), gvel (N, d, g...); tables as for advchain_tp_interp_fwd (3 axes, trivial leading
 * one), wd / wlo / WB: the densified innermost bands of advchain_band_reduce_rows_dense.  Returns ADVCHAIN_ERR_UNSUPPORTED (-2)
 * with NOTHING enqueued when a launch would not take the shape: issue the separate calls then.                              */
int advchain_demons_compose_pair_fwd(const float* vel, float* s1, float* phi0, float* fields, float* pos, float* q,
                                     float* disp, float* rows_max, const int32_t* itab, const float* ftab,
                                     const int64_t* S3, const int64_t* g3, const int64_t* B3, int64_t N, int ndim, int n,
                                     const int32_t* hints, const float* weights9_host, float scale, float inv, int fuse,
                                     void* stream);
int advchain_demons_compose_pair_bwd(const float* gq_lo, const float* gq_hi, const float* pos, const float* phi0,
                                     const float* fields, float* gpos, float* g, float* scratch, int32_t* ws, float* t1,
                                     float* gs1, float* gvel, const int32_t* halos, const int32_t* itab, const float* ftab,
                                     const float* wd, const int32_t* wlo, int64_t WB, const int64_t* S3, const int64_t* g3,
                                     const int64_t* B3, int64_t N, int ndim, int n, const float* weights9_host, float scale,
                                     float inv, void* stream);
/* max over samples and axes of |sampling position - own voxel| of the field phi, in voxels: the displacement
 * bound `halo` above wants.  out: one float, zero-initialised by the caller (atomic max).            */
int advchain_max_displacement(const float* phi, float* out, int64_t N, int ndim, const int64_t* dims, void* stream);
/* out[r] = max over the `cols` accumulator slots of row r (replaces torch.max(dim=1) on the (squarings + 1) x
 * ADVCHAIN_DISP_SLOTS tensor the squaring chain fills: 3 us instead of a 13-us two-pass reduction, 12 times per call);
 * NaN propagates.                                                                                        */
/* reset != 0: the slots are zeroed after they are read, so that one persistent accumulator serves every chain (no
 * zero-fill launch per chain). */
int advchain_slot_rows_max(float* slots, float* out, int64_t rows, int64_t cols, int reset, void* stream);
/* Premise check of a replayed launch plan (hipGraph replay of a solver call, INTEGRATION.md "Graph replay"): the reference
 * has no counterpart -- its ATen call sites (adv_morph.py:116-202, 546-557) do not choose between formulations.  The backward
 * kernels above are chosen from displacement bounds; a replayed plan froze that choice, and this entry compares what the
 * replay measures (`values`: advchain_slot_rows_max / advchain_max_displacement output, or the 3D step-rule norm of
 * adv_morph.py:159-162) with the frozen interval: flag[0] |= 1 when some values[i] is outside [lo[i], hi[i]) or NaN.  */
int advchain_bounds_check(const float* values, const float* lo, const float* hi, int64_t n, int32_t* flag, void* stream);

/* ---- affine warp -----------------------------------------------------------------------
 * replaces: F.affine_grid(theta, size, align_corners=True) + F.grid_sample(...),
 *           adv_affine.py:297-313.  theta (N, ndim, ndim+1); the grid is never materialised.    */
int advchain_affine_warp_fwd(const float* in, const float* theta, float* out, int64_t N, int64_t C, int ndim,
                             const int64_t* dims, int interp, int padding, void* stream);
/* advchain_affine_warp_fwd for two tensors under ONE theta: out = warp(in), ride_out = warp(ride_in) (one channel); see
 * advchain_grid_sample_fwd_ride (adv_affine.py:297-313 called for the data and for the validity mask,
 * adv_compose_solver.py:262-268).  Linear / zeros / rows of 16 bytes: one launch in 2D and 3D; otherwise two.         */
int advchain_affine_warp_fwd_ride(const float* in, const float* theta, float* out, const float* ride_in, float* ride_out,
                                  int64_t N, int64_t C, int ndim, const int64_t* dims, int interp, int padding, int flags,
                                  void* stream);
int64_t advchain_affine_warp_bwd_workspace(int64_t N, int ndim, const int64_t* dims); /* floats */
/* grad_in (N,C,dims) and grad_theta (N, ndim, ndim+1) are overwritten; either may be NULL.  grad_theta uses a
 * deterministic two-stage reduction.  grad_in: for linear interpolation with zeros padding and C <= 4 it is an
 * owner-computes scatter into LDS fixed-point accumulators (no global atomics, deterministic; scaled by the largest
 * |grad_out| around the tile -- taken from the theta-gradient walk, which runs first when both are asked for), for
 * C <= 8 a gather over the affine lattice; otherwise it is zero-filled here and scattered with atomics.
 * `workspace` (advchain_affine_warp_bwd_workspace floats) is always required.                                   */
int advchain_affine_warp_bwd(const float* grad_out, const float* in, const float* theta, float* grad_in,
                             float* grad_theta, float* workspace, int64_t N, int64_t C, int ndim,
                             const int64_t* dims, int interp, int padding, void* stream);

/* ---- bicubic sampling (2D) ----------------------------------------------------------------
 * replaces: F.grid_sample(data, grid, mode='bicubic', padding_mode, align_corners=True), reached through the
 *           forward_interp / backward_interp config keys (adv_morph.py:255-258,546-557; adv_affine.py:297-313; ATen has
 *           no 5-D bicubic: 2D only).  Cubic convolution, A = -0.75, every tap's coordinate padded on its own.
 * in (N,C,H,W), grid (N,2,OH,OW) planar, out (N,C,OH,OW).  bwd: grad_in is zero-filled here and accumulated with
 * atomics, grad_grid overwritten; either may be NULL.                                                              */
int advchain_grid_sample_bicubic2d_fwd(const float* in, const float* grid, float* out, int64_t N, int64_t C,
                                       const int64_t* in_dims, const int64_t* out_dims, int padding, void* stream);
int advchain_grid_sample_bicubic2d_bwd(const float* grad_out, const float* in, const float* grid, float* grad_in,
                                       float* grad_grid, int64_t N, int64_t C, const int64_t* in_dims,
                                       const int64_t* out_dims, int padding, void* stream);
/* replaces: F.affine_grid(theta, size, align_corners=True) for the bicubic affine warp (adv_affine.py:297-305): theta
 *           (N,2,3) -> planar grid (N,2,H,W), and its adjoint grad_grid -> grad_theta (deterministic two-stage sum;
 *           workspace: advchain_affine_grid2d_bwd_workspace floats).                                                */
int advchain_affine_grid2d_fwd(const float* theta, float* grid, int64_t N, const int64_t* dims, void* stream);
int64_t advchain_affine_grid2d_bwd_workspace(int64_t N, const int64_t* dims);
int advchain_affine_grid2d_bwd(const float* grad_grid, float* grad_theta, float* workspace, int64_t N, const int64_t* dims,
                               void* stream);

/* ---- affine parameters -> matrices -----------------------------------------------------
 * replaces: AdvAffine.gen_batch_affine_matrix adv_affine.py:210-273 (Hardtanh, Euler z-y'-x''
 *           rotation, scale, translation) and get_inverse_matrix adv_affine.py:316-324.
 * param (N, 5|9); cfg = 2D {rot, scale_x, scale_y, shift_x, shift_y}
 *                       3D {rot_x,rot_y,rot_z, scale_x,scale_y,scale_z, shift_x,shift_y,shift_z};
 * theta, theta_inv (N, ndim, ndim+1).                                                         */
int advchain_affine_theta_fwd(const float* param, const float* cfg_host, float param_scale, float* theta,
                              float* theta_inv, int64_t N, int ndim, void* stream);
int advchain_affine_theta_bwd(const float* param, const float* cfg_host, float param_scale, const float* grad_theta,
                              const float* grad_theta_inv, float* grad_param, int64_t N, int ndim, void* stream);

/* ---- banded tensor-product interpolation -----------------------------------------------
 * Band tables (see advchain_amd/bands.py) describe, per axis a of the padded 3-axis view
 * (2D passes a trivial leading axis), a linear map from g_a coefficients to S_a samples with at
 * most B_a (<= 8) contiguous non-zeros per sample:
 *   itab = for a in 0..2: start[S_a] | lo[g_a] | hi[g_a]      (int32)
 *   ftab = for a in 0..2: w[S_a * B_a]                        (fp32)
 * replaces: F.interpolate(duv, size=full, 'bilinear'|'trilinear', align_corners=False)
 *           adv_morph.py:464, fused with 'basegrid += duv/2^n' (adv_morph.py:111,129-130) when
 *           add_identity != 0, and with torch.norm(duv_interval) (adv_morph.py:160) when
 *           sumsq != NULL (64 partial accumulators: sum(sumsq[0..63]) += sum(interp^2); caller zeroes them).
 * coef (planes, g0,g1,g2) -> out (planes, S0,S1,S2) = identity? + scale * interp.
 * disp_out (may be NULL): ADVCHAIN_DISP_SLOTS floats, max-accumulates |scale * interp| in voxels, i.e. the displacement
 * of the field written when add_identity != 0 (see advchain_compose_self_fwd).                          */
int advchain_tp_interp_fwd(const float* coef, float* out, const int32_t* itab, const float* ftab, const int64_t* S,
                           const int64_t* g, const int64_t* B, int64_t planes, int64_t C, int ndim, int add_identity,
                           float scale, float* sumsq, float* disp_out, void* stream);
/* advchain_gauss_small_pair (forward) + advchain_tp_interp_fwd in ONE launch (round 6) for the paired 2D field [v; -v].
 * replaces: the first two steps of AdvMorph.DemonsCompose -- the Gaussian of the low-resolution velocity (adv_morph.py:377-452,
 *           called at :460-462) and F.interpolate to full size (adv_morph.py:464) -- for forward() / backward() of one solver
 *           step (adv_morph.py:299-303,322-324).  vel: (P planes of g1 x g2 values); s1 (may be NULL): (2P planes) receives the
 *           smoothed batch [G(gscale v); G(-gscale v)]; out: (2P planes of S1 x S2) = (add_identity ? identity : 0) + scale * up(s1);
 *           disp_out as for advchain_tp_interp_fwd.  Same arithmetic as the two calls, bit for bit.  2D tables (trivial leading
 *           axis), low-resolution planes of at most 1024 values, the 9-tap window; ADVCHAIN_ERR_UNSUPPORTED (-2) with nothing
 *           enqueued otherwise.                                                                                              */
int advchain_tp_interp_fwd_smoothed_pair(const float* vel, float* s1, float* out, const int32_t* itab, const float* ftab,
                                         const int64_t* S, const int64_t* g, const int64_t* B, int64_t P, int64_t C,
                                         int add_identity, float scale, float* disp_out, const float* weights9, float gscale,
                                         void* stream);
/* adjoint along one axis: in (outer, S_axis, inner) -> out (outer, g_axis, inner),
 * out = W_axis^T ((in - in2) * scale); in2 may be NULL.
 * replaces: upsample_{bi,tri}linear backward / conv_transpose backward w.r.t. its input.        */
int advchain_band_reduce_axis(const float* in, const float* in2, float* out, const int32_t* itab, const float* ftab,
                              const int64_t* S, const int64_t* g, const int64_t* B, int axis, int64_t outer,
                              int64_t inner, float scale, void* stream);
/* The innermost-axis pass of the same adjoint (inner == 1: the full-resolution pass, which reads the whole gradient once) with
 * the bands densified by the caller once per table: wd[g][WB] = weight of input lo[k] + j for coefficient k (zero beyond its
 * band), lo[g].  in: (rows, S) [minus in2] -> out: (rows, g) * scale.  Returns ADVCHAIN_ERR_UNSUPPORTED (-2) for shapes it
 * does not take (S % 4 != 0, S > 1024, g > 64, g * WB > 4096, unaligned inputs): use advchain_band_reduce_axis then.
 * Same sums in the same order as that entry. */
int advchain_band_reduce_rows_dense(const float* in, const float* in2, float* out, const float* wd, const int32_t* lo,
                                    int64_t rows, int64_t S, int64_t g, int64_t WB, float scale, void* stream);

/* ---- bias field ------------------------------------------------------------------------
 * replaces: AdvBias.compute_smoothed_bias + clip_bias + multiply, adv_bias.py:279-356,186:
 *           conv_transpose{2,3}d(cp, bspline) -> crop -> Upsample(linear, align_corners=False)
 *           -> exp (log space) or 1+x -> 1+clamp(b-1,-eps,eps) -> data*b, evaluated in closed
 *           form from the control points (band tables = upsample o B-spline, per axis).
 * cp (N,1,g...), data/out (N,C,S...), field (N,1,S...) = clipped bias.  data may be NULL.        */
int advchain_bias_field_fwd(const float* cp, const float* data, float* out, float* field, const int32_t* itab,
                            const float* ftab, const int64_t* S, const int64_t* g, const int64_t* B, int64_t N,
                            int64_t C, float eps, int use_log, float cp_scale, void* stream);
/* grad_L (N,1,S...) = dLoss/d(log-field) at full resolution (reduce to control points with
 * advchain_band_reduce_axis); grad_data (N,C,S...) = grad_out * field.  Either may be NULL.     */
int advchain_bias_field_bwd(const float* cp, const float* data, const float* grad_out, float* grad_L, float* grad_data,
                            const int32_t* itab, const float* ftab, const int64_t* S, const int64_t* g,
                            const int64_t* B, int64_t N, int64_t C, float eps, int use_log, float cp_scale,
                            void* stream);

/* ---- separable Gaussian ----------------------------------------------------------------
 * replaces: depthwise nn.Conv{2,3}d with the normalised 9^d Gaussian (sigma=1), zero padding,
 *           adv_morph.py:377-452; one axis per call (self-adjoint: same entry for backward).
 * pre : 0 none | 1 x*scale | 2 F.grid_sample(base_grid, x, 'border') - base_grid  (adv_morph.py:473-487)
 * post: 0 none | 1 + base_grid (adv_morph.py:489)   | 2 * d(pre 2)/dx at aux      (backward of pre 2)
 * in/out/aux (planes, dims); plane p carries grid channel p % C.  host pointer weights9[9].     */
int advchain_gauss_axis(const float* in, float* out, const float* aux, int64_t planes, int64_t C, int ndim,
                        const int64_t* dims, int axis, const float* weights9_host, int pre, int post, float scale,
                        void* stream);

/* The same convolution with ANY odd window (host pointer weights[ntaps], 1 <= ntaps <= 129): the reference sizes its window
 * as 2 * int(4 sigma + 0.5) + 1 taps (adv_morph.py:393-398), 9 only for 0.875 <= sigma < 1.125; out = G_axis * (scale * in),
 * zero padding, no prologue / epilogue.  Plain per-voxel form (the reference never leaves sigma = 1).                       */
int advchain_gauss_axis_generic(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims, int axis,
                                const float* weights_host, int ntaps, float scale, void* stream);

/* x and y passes of the Gaussian above in ONE launch (LDS tile of whole rows): one read and one write of the tensor
 * instead of two each; the same sums in the same tap order as the two per-axis calls (results agree to a few ulp).  post != 0 only when y is the last axis (ndim == 2).
 * Returns -2 (unsupported, no error text) for shapes it does not take (rows not a multiple of 4 or longer than 512, unaligned
 * tensors): run advchain_gauss_axis for axes 2 and 1 then.
 * in_hi (may be NULL): the input planes [planes_lo, planes) come from this second tensor (the gradient of a paired field
 * arrives as two halves), in: planes [0, planes_lo).                                                                    */
int advchain_gauss_xy(const float* in, float* out, const float* aux, int64_t planes, int64_t C, int ndim, const int64_t* dims,
                      const float* weights9, int pre, int post, float scale, void* stream, const float* in_hi,
                      int64_t planes_lo);
/* all axes of the same Gaussian in ONE launch for small planes (<= 4096 voxels: the low-resolution velocity grids,
 * adv_morph.py:462-463); pre = 0 | 1 (x * scale), no epilogue; same arithmetic as the per-axis calls.              */
int advchain_gauss_small(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims,
                         const float* weights9_host, int pre, float scale, void* stream);

/* The same for the batch [v; -v] of a paired field (the deformation and its approximate inverse of one solver step,
 * adv_morph.py:285-331) without materialising the negated copy: adjoint == 0: in (planes), out (2 planes): out[p] =
 * G(scale in[p]), out[planes + p] = G(-scale in[p]) (bit-identical to smoothing a negated copy); adjoint != 0: in
 * (2 planes), out (planes): out[p] = G(scale in[p]) - G(scale in[planes + p]).                                       */
int advchain_gauss_small_pair(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims,
                              const float* weights9_host, float scale, int adjoint, void* stream);

/* ---- streaming / per-sample normalisation ----------------------------------------------
 * replaces: data + eps*param adv_noise.py:81-84 (x may be NULL: out = a*y).                     */
int advchain_axpy(const float* x, const float* y, float* out, float a, int64_t n, void* stream);
/* replaces: param + step * grad.sign() (adv_affine.py:186-195); base may be NULL (power iteration: sign only).   */
/* gate (may be NULL): a device scalar (the loss of the step); when it is NaN / inf the update is void and out = old --
 * the NaN guard of the ascent loop (adv_compose_solver.py:343-347) evaluated on the device, so that the host does not have
 * to read the loss back in the middle of a step.                                                                      */
int advchain_sign_axpy(const float* base, const float* x, float* out, float a, int64_t n, const float* gate, const float* old,
                       void* stream);
/* replaces: fb[fb != 0] = 1 of the validity mask (adv_compose_solver.py:266-268,323-325): out = (x != 0) ? 1 : 0.  */
int advchain_nonzero_mask(const float* x, float* out, int64_t n, void* stream);
int64_t advchain_norm_workspace(int64_t N, int64_t M); /* floats */
/* replaces: unit_normalize (adv_transformation_base.py:151-155) fused with the ascent update
 *           param + step*g/(||g||+1e-20) (adv_noise.py:56-63, adv_bias.py:144-147,
 *           adv_morph.py:511-513).  base may be NULL (pure normalisation).  x (N, M).           */
int advchain_norm_axpy(const float* base, const float* x, float* out, float* workspace, float step, int64_t N,
                       int64_t M, void* stream);
/* the same with the NaN gate of advchain_sign_axpy */
int advchain_norm_axpy_gated(const float* base, const float* x, float* out, float* workspace, float step, int64_t N,
                             int64_t M, const float* gate, const float* old, void* stream);
/* The parameter updates of ONE ascent step in ONE launch (round 6).
 * replaces: the per-transform optimize_parameters() calls at the end of a step (adv_compose_solver.py:349-364 ->
 *           adv_noise.py:51-64, adv_bias.py:139-148, adv_morph.py:501-516: param + step * unit_normalize(grad);
 *           adv_affine.py:182-198: param + step * sign(grad)) and the NaN guard of adv_compose_solver.py:343-347 -- what
 *           advchain_norm_axpy_gated / advchain_sign_axpy do one transform at a time.  descs: n <= 8 HOST descriptors;
 *           kind 0: out[r] = (base ? base[r] : 0) + step * x[r] / (||x[r]||_2 + 1e-20) per row r of M values; kind 1: the sign
 *           step.  gate (may be NULL): a device scalar; when it is NaN / inf every `out` receives `old` (required then).
 *           One workgroup per row: same formula as the separate entries, square sums in a different order (rounding).   */
typedef struct advchain_update_desc {
  const float* base;
  const float* x;
  float* out;
  const float* old;
  int64_t N;
  int64_t M;
  int32_t kind;
  float step;
} advchain_update_desc;
int advchain_update_multi(const advchain_update_desc* descs, int n, const float* gate, void* stream);

/* ---- consistency loss ------------------------------------------------------------------
 * replaces: calc_segmentation_consistency / contour_loss / kl_divergence, advchain/common/loss.py:8-87,102-220,223-249
 *           ('mse', 'contour' and 'kl' terms: softmax over K, mask, 3^d edge stencils, Q13/Q14).
 * pred/ref (N,K,dims) logits (ref already a probability map when ref_is_prob), mask
 * (N, mask_channels in {1,K}, dims) or NULL.  Outputs: P = softmax(pred), D = P - T (N,K,dims);
 * R (N, 2(K-1), dims) = 2 m^2 (A*D), 2 m^2 (B*D) per class 1..K-1 (NULL when no backward is
 * needed); sums is 4 x 64 partial accumulators (row r, slot s at sums[64 r + s]; per-workgroup partials are
 * spread over the slots because same-address atomics serialise): row 0 += sum ((P m)-(T m))^2, row 1 +=
 * sum (A*D m)^2, row 2 += sum (B*D m)^2, row 3 (want_kl != 0 only) += sum_k m_k T'_k (log T'_k - log P_k) with
 * T' = T, or where(ref == 0, 1e-8, 1 - 1e-8) when ref_is_prob (loss.py:239-243) (caller zeroes `sums`, adds the slots up
 * and applies the GLOBAL normalisers -- required for batch sharding).  2D: A = Sobel-x, B = Sobel-y.
 * 3D: A = h(x)hp(x)h (the reference uses it for conv_x AND conv_y), B = h(x)h(x)hp.                */
int advchain_consistency_fwd(const float* pred, const float* ref, const float* mask, float* P, float* D, float* R,
                             float* sums, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                             int ref_is_prob, int want_edges, int want_kl, void* stream);
/* replaces: the sums and the weighted total of calc_segmentation_consistency (loss.py:80-87): sums[r] = sum of the 64
 * slots of row r (4 floats), value[0] = sum_r coef4[r] * sums[r]; reset != 0 zeroes the slots for the next call.  */
int advchain_consistency_finish(float* slots, const float* coef4_host, float* sums, float* value, int reset, void* stream);
/* grad_pred (N,K,dims) = softmax'(P)[ gs (c_mse 2 m^2 D + c_a A^T R_A + c_b B^T R_B) ]
 *                        + gs c_kl (P_j sum_k m_k T'_k - m_j T'_j),   gs = *grad_scale (device scalar, NULL = 1);
 * the 'kl' term (c_kl != 0) rebuilds T' from P - D (kl_is_gt: the where() of the forward).        */
int advchain_consistency_bwd(const float* P, const float* D, const float* R, const float* mask,
                             const float* grad_scale, float* grad_pred, float c_mse, float c_a, float c_b, float c_kl,
                             int kl_is_gt, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                             void* stream);
/* f2, fused (round 4): the same loss without its per-voxel intermediates.  fused_fwd: softmax(pred), softmax(ref) (or ref
 * itself), the masked squared error, the KL sum and the 3^d edge stencils in ONE marching kernel straight from the logits;
 * it writes only R (2(K-1) channels; may be NULL when no gradient is wanted) and adds to the same 4 x 64 slot sums as
 * advchain_consistency_fwd (finish with advchain_consistency_finish).  fused_bwd: grad_pred from pred, ref, R -- the softmax is
 * recomputed at the voxel instead of reading P and D back.  Same per-voxel arithmetic as the unfused entries (replaces the
 * same reference lines: common/loss.py:8-87,102-220,223-249).  Both return ADVCHAIN_ERR_UNSUPPORTED (-2) for what the 16-byte
 * marching form does not take -- rows of 4j <= 256 voxels, K = 2..4, a mask of at most one channel, 16-byte aligned
 * tensors: use the unfused entries then.  3D (round 5): with the edge terms wanted and rows of at most 128 voxels the march goes
 * along z with the y neighbours of a row exchanged through LDS (every logit is read ~1.4 x instead of 3.75 x); other 3D calls
 * return -2 as before. */
int advchain_consistency_fused_fwd(const float* pred, const float* ref, const float* mask, float* R, float* sums, int64_t N,
                                   int64_t K, int ndim, const int64_t* dims, int mask_channels, int ref_is_prob, int want_edges,
                                   int want_kl, void* stream);
int advchain_consistency_fused_bwd(const float* pred, const float* ref, const float* R, const float* mask,
                                   const float* grad_scale, float* grad_pred, float c_mse, float c_a, float c_b, float c_kl,
                                   int ref_is_prob, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                                   void* stream);

/* bf16 STORAGE experiment (round 6; BASELINE config 2 names "bf16"): the 2D K = 4 fused loss above (common/loss.py:8-87,
 * 102-220: mse + contour terms on logits) with pred / ref / R / grad_pred stored as bfloat16 (raw 16-bit words, 8-byte aligned)
 * and all arithmetic in fp32 registers.  NOT used by the product path -- the parity contract is fp32 at 1e-4; the entries exist
 * so that "what would half the bytes buy" is a measurement (tools/kernel_bench.py).  mask: fp32, one
 * channel or NULL.  ADVCHAIN_ERR_UNSUPPORTED (-2) for anything but ndim == 2, K == 4, rows of 4j <= 256 pixels. */
int advchain_consistency_fused_fwd_bf16(const void* pred, const void* ref, const float* mask, void* R, float* sums, int64_t N,
                                        int64_t K, int ndim, const int64_t* dims, void* stream);
int advchain_consistency_fused_bwd_bf16(const void* pred, const void* ref, const void* R, const float* mask,
                                        const float* grad_scale, void* grad_pred, float c_mse, float c_a, float c_b, int64_t N,
                                        int64_t K, int ndim, const int64_t* dims, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVCHAIN_HIP_H_ */
