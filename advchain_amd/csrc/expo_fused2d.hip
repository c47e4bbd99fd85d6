// Fused early squarings of the 2D scaling-and-squaring chain (gfx950): phi_0 -> phi_1 .. phi_k in ONE launch.
//
//   replaces k launches of advchain_compose_self_fwd at the start of vectorFieldExponentiation2D
//   (adv_morph.py:116-146: `for i in range(nb_steps): phi = applyComposition2D(phi, phi)`, each a
//   F.grid_sample(phi, phi^T, padding_mode='border', align_corners=True), adv_morph.py:179-190)
//
// Squaring m composes a field that moves a sample by 2^(m-n) of the final displacement: the first squarings of a chain
// move every sample by less than a pixel.  A sample of phi_j then takes its four corners from phi_{j-1} within +-1 row, so
// a workgroup that owns TH whole rows and stages rows y0-k .. y0+TH+k-1 of phi_0 in LDS can produce phi_1 on that window
// shrunk by one row either side, phi_2 shrunk by two, ..., phi_k on its own rows: k squarings for one read of the field
// (1 + 2k/TH of it) and k writes, k-1 launches and k-1 re-reads of a 33-MB field less (cfg-2: 64 x 2 x 256 x 256).
//
//   * whole rows: no halo in x (border padding clips at the row ends), every global access is a 16-byte quad of a row;
//   * LDS holds the window as interleaved (x, y) pairs: the two x corners of a row are ONE ds_read2_b64, the four corners
//     of a sample two; lane <-> consecutive pixel for the taps (bank-conflict free), thread <-> quad for loads / stores;
//   * one window buffer, updated in place: results wait in registers across a barrier (2 barriers per level);
//   * the arithmetic is that of k_compose_self_fwd<2, .> (Taps<2, PAD_BORDER>, the paired-corner select of CornerOffsets,
//     the fma chain of sample_linear<2>): the fields are BIT-IDENTICAL to the unfused launches (tests/test_fused2d_gpu.py);
//   * the sub-pixel premise is CHECKED, not assumed, level by level: a workgroup knows the displacement of what its window
//     currently holds (measured while staging phi_0, then on every level's results) and runs the next level only if it is
//     below 0.999 pixel -- every corner of that level then lies in the rows the previous level produced.  A workgroup that
//     has to stop after j < k levels records the deficit k - j in `fail_flag` (a float, atomic max over the workgroups; 0 =
//     every workgroup did all k levels).  The chain enqueues ONE repeat launch behind this kernel (k_expo_repeat2d,
//     sampler.hip): it returns at once while the flag is down; otherwise it runs the levels some window could not do the
//     ordinary way on every pixel (a persistent grid with a grid barrier between levels), which rewrites identical bits
//     where the fused kernel did get that far.
#include "sampler_common.h"

namespace advchain {

constexpr int kFuseMaxLevels = 5;

// NT threads; a wave handles PPW segments of 64 consecutive pixels of the window (segment s of the window = row
// s / (W / 64), columns 64 * (s % (W / 64)) ..); requires W % 64 == 0.
template <int NT, int PPW>
__global__ void __launch_bounds__(NT)      // (101 VGPRs, two workgroups a CU; capped at 80 for three it spills and is 4 % slower)
k_expo_fused_fwd2d(const float* __restrict__ phi0, float* __restrict__ fields, int64_t F, Dims d, int k, int TH, int hal,
                   float* __restrict__ disp_rows, float* __restrict__ fail_flag) {
  extern __shared__ float2 win[];            // [WY][W] (x, y) of the current level
  __shared__ float red[NT / 64];
  const int W = d.s2, S1 = d.s1;
  const int V = W * S1;
  const int n = blockIdx.y;
  const int y0 = blockIdx.x * TH;
  // hal: 4 bits per level, the row halo h of level lev = a bound on the displacement of ITS INPUT in pixels (1 for the
  // sub-pixel levels); the window carries their sum either side
  int HS = 0;
  for (int j = 0; j < k; ++j) HS += (hal >> (4 * j)) & 15;
  const int wy0 = y0 - HS;                   // image row of window row 0
  const int WY = TH + 2 * HS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NWV = NT / 64;
  const int SPR = W >> 6;                    // segments per row
  const int NSEG = WY * SPR;
  const float* pn = phi0 + (int64_t)n * 2 * V;
  // window rows that exist in the image (corner rows are clamped into them for memory safety; under the displacement
  // premise they lie there anyway)
  const int glo = max(wy0, 0), ghi = min(wy0 + WY, S1) - 1;

  // ---- stage phi_0: thread <-> quad, 16-byte loads per channel, interleaved into LDS; measure the window's displacement
  const int QW = W >> 2;
  float dloc = 0.f;
  for (int q = threadIdx.x; q < WY * QW; q += NT) {
    const int wr = q / QW, qx = q - wr * QW;
    const int gy = wy0 + wr;
    const bool in_img = gy >= 0 && gy < S1;
    const int gyc = min(max(gy, 0), S1 - 1);
    const float4 vx = *reinterpret_cast<const float4*>(pn + gyc * W + 4 * qx);
    const float4 vy = *reinterpret_cast<const float4*>(pn + V + gyc * W + 4 * qx);
    float4* dst = reinterpret_cast<float4*>(win + wr * W + 4 * qx);
    dst[0] = make_float4(vx.x, vy.x, vx.y, vy.y);
    dst[1] = make_float4(vx.z, vy.z, vx.w, vy.w);
    if (in_img) {
      const float xs[4] = {vx.x, vx.y, vx.z, vx.w}, ys[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (NaN compares false in fmaxf's favour: a NaN field must not pass the test)
        const float dx = voxel_displacement(xs[j], W, 4 * qx + j), dy = voxel_displacement(ys[j], S1, gy);
        dloc = (dx == dx && dy == dy) ? fmaxf(dloc, fmaxf(dx, dy)) : 1.0e9f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dloc = fmaxf(dloc, __shfl_xor(dloc, o, 64));
  if (lane == 0) red[wave] = dloc;
  __syncthreads();
  float dcur = 0.f;                          // displacement of what the window holds, over the rows the next level reads
#pragma unroll
  for (int w = 0; w < NWV; ++w) dcur = fmaxf(dcur, red[w]);

  // the segments of this wave: window row and first column (wave-uniform, computed once)
  int srow[PPW], scol[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int seg = wave + i * NWV;
    srow[i] = seg < NSEG ? seg / SPR : -1;
    scol[i] = (seg - (seg / SPR) * SPR) << 6;
  }

  const float topx = (float)(W - 1), topy = (float)(S1 - 1), hx = 0.5f * topx, hy = 0.5f * topy;
  int cum = 0;                                 // rows the window has shrunk by either side
  for (int lev = 1; lev <= k; ++lev) {
    const int h = (hal >> (4 * (lev - 1))) & 15;
    if (!(dcur < (h == 1 ? 0.999f : (float)h - 0.001f))) {   // block-uniform (NaN included): this window cannot do level `lev`
      if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(fail_flag), __float_as_uint((float)(k - lev + 1)));
      return;
    }
    // rows this level produces: the window shrunk by the halos so far either side, inside the image
    cum += h;
    const int rlo = max(wy0 + cum, 0), rhi = min(wy0 + WY - cum, S1) - 1;
    float rx[PPW], ry[PPW];
    float dmax = 0.f, dall = 0.f;              // displacement of phi_lev over the owned rows / over every row of the level
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int wr = srow[i];                         // wave-uniform
      const int gy = wy0 + wr;
      rx[i] = ry[i] = 0.f;
      if (wr < 0 || gy < rlo || gy > rhi) continue;
      const int px = scol[i] + lane;
      const float2 own = win[wr * W + px];
      // The taps of k_compose_self_fwd<2, .> (make_tap<PAD_BORDER> + sample_linear<2>), minus what the premise makes dead:
      // the window is finite (a NaN / inf raises the flag above), so the position is a finite number clipped into
      // [0, S - 1], the lower corner is always inside, and the upper corner is outside only where its weight is exactly 0
      // -- `ok ? v : 0` then adds the same +-0 as v * 0 with a finite v.  Same operations, same order, same roundings.
      // ((c + 1) * 0.5) * (S - 1) == (c + 1) * (0.5 (S - 1)) bit for bit: the halving is exact
      const float xs = __builtin_amdgcn_fmed3f((own.x + 1.f) * hx, 0.f, topx);
      const float ys = __builtin_amdgcn_fmed3f((own.y + 1.f) * hy, 0.f, topy);
      const float fx = floorf(xs), fy = floorf(ys);
      const int ix = (int)fx, iy = (int)fy;
      const float wx1 = xs - fx, wx0 = (fx + 1.f) - xs, vy1 = ys - fy, vy0 = (fy + 1.f) - ys;   // (vy: the y weights)
      // the two x corners of a row are one 16-byte pair starting at xa = min(ix, W - 2) (CornerOffsets' rule).  Only for
      // ix == W - 1 (position exactly on the right border, wx1 == 0) the pair is (W - 2, W - 1) instead of (W - 1, W): its
      // first element then carries weight 0 and its second the lower corner's weight -- the same non-zero terms in the
      // same order as the per-corner form
      const int xa = min(ix, W - 2);
      const bool edge = ix != xa;
      const float wa = edge ? 0.f : wx0, wb = edge ? wx0 : wx1;
      const int r0 = min(max(iy, glo), ghi) - wy0, r1 = min(max(iy + 1, glo), ghi) - wy0;
      const float2 a0 = win[r0 * W + xa], b0 = win[r0 * W + xa + 1];
      const float2 a1 = win[r1 * W + xa], b1 = win[r1 * W + xa + 1];
      const float w00 = wa * vy0, w01 = wb * vy0, w10 = wa * vy1, w11 = wb * vy1;
      float ax = tap_acc<2>(0.f, a0.x, w00), ay = tap_acc<2>(0.f, a0.y, w00);
      ax = tap_acc<2>(ax, b0.x, w01); ay = tap_acc<2>(ay, b0.y, w01);
      ax = tap_acc<2>(ax, a1.x, w10); ay = tap_acc<2>(ay, a1.y, w10);
      ax = tap_acc<2>(ax, b1.x, w11); ay = tap_acc<2>(ay, b1.y, w11);
      rx[i] = ax; ry[i] = ay;
      const float dpx = fmaxf(fabsf(((ax + 1.f) * 0.5f) * (float)(W - 1) - (float)px),          // voxel_displacement() of a
                              fabsf(((ay + 1.f) * 0.5f) * (float)(S1 - 1) - (float)gy));        // finite value
      dall = fmaxf(dall, dpx);
      if (gy >= y0 && gy < y0 + TH) dmax = fmaxf(dmax, dpx);      // ... over the rows this workgroup owns
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dall = fmaxf(dall, __shfl_xor(dall, o, 64));
    if (lane == 0) red[wave] = dall;
    __syncthreads();                                  // every tap of this level has been read
    dcur = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) dcur = fmaxf(dcur, red[w]);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int wr = srow[i];
      const int gy = wy0 + wr;
      if (wr < 0 || gy < rlo || gy > rhi) continue;
      win[wr * W + scol[i] + lane] = make_float2(rx[i], ry[i]);
    }
    if (disp_rows) wave_max_to_slots(dmax, disp_rows + (int64_t)lev * kDispSlots);
    __syncthreads();                                  // phi_lev is in the window
    // ---- phi_lev of the owned rows -> fields[lev - 1]: thread <-> quad, 16-byte stores per channel
    float* on = fields + (int64_t)(lev - 1) * F + (int64_t)n * 2 * V;
    const int own_rows = min(TH, S1 - y0);
    for (int q = threadIdx.x; q < own_rows * QW; q += NT) {
      const int r = q / QW, qx = q - r * QW;
      const float4* src = reinterpret_cast<const float4*>(win + (r + HS) * W + 4 * qx);
      const float4 p0 = src[0], p1 = src[1];
      *reinterpret_cast<float4*>(on + (y0 + r) * W + 4 * qx) = make_float4(p0.x, p0.z, p1.x, p1.z);
      *reinterpret_cast<float4*>(on + V + (y0 + r) * W + 4 * qx) = make_float4(p0.y, p0.w, p1.y, p1.w);
    }
  }
}

}  // namespace advchain

using namespace advchain;

// phi_1..phi_k of a 2D chain in one launch (k >= 1).  `halos`: 4 bits per level, a bound (pixels, 1..15) on the displacement
// of the level's input -- 1 for the sub-pixel squarings, larger for the squarings behind them (the window then carries
// the sum of the halos either side).  ADVCHAIN_ERR_UNSUPPORTED when the shape does not fit (rows must be a multiple of 64
// pixels and at most 512, 16-byte aligned base pointers, a window of at most 64 KiB); otherwise the launch is enqueued and
// `fail_flag` (one float, zero before the call) is raised by any workgroup whose window moves too far for one of its levels.
// `query`: nothing is enqueued -- the return value says whether the shape would be taken (pointers are not looked at).
int advchain_expo_fused_fwd2d_launch(const float* phi0, float* fields, int64_t N, Dims d, int k, int halos, float* disp_rows,
                                     float* fail_flag, hipStream_t stream, bool query) {
  if (d.s0 != 1 || k < 1 || k > kFuseMaxLevels || (!fail_flag && !query)) return ADVCHAIN_ERR_UNSUPPORTED;
  const int W = d.s2;
  if (W % 64 != 0 || W > 512 || d.s1 < 8) return ADVCHAIN_ERR_UNSUPPORTED;
  if (!query && ((reinterpret_cast<uintptr_t>(phi0) | reinterpret_cast<uintptr_t>(fields)) & 15)) return ADVCHAIN_ERR_UNSUPPORTED;
  int HS = 0;
  for (int j = 0; j < k; ++j) {
    const int h = (halos >> (4 * j)) & 15;
    if (h < 1) return ADVCHAIN_ERR_UNSUPPORTED;
    HS += h;
  }
  constexpr int NT = 512, PPW = 13;      // (16 segments a wave: 133 VGPRs, one workgroup a CU instead of two)
  // rows per workgroup: the window (TH + 2 HS rows of W pairs) within 56 KiB (two workgroups a CU) and within the PPW
  // segments a wave can carry; 16 where that fits
  const int spr = W / 64;
  int TH = 16;
  while (TH > 4 && ((TH + 2 * HS) * spr > PPW * (NT / 64) || (size_t)(TH + 2 * HS) * W * sizeof(float2) > 56 * 1024)) TH -= 4;
  if ((TH + 2 * HS) * spr > PPW * (NT / 64) || (size_t)(TH + 2 * HS) * W * sizeof(float2) > 64 * 1024) return ADVCHAIN_ERR_UNSUPPORTED;
  const int64_t F = N * 2 * d.voxels();
  const size_t lds = (size_t)(TH + 2 * HS) * W * sizeof(float2);
  dim3 grid((unsigned)((d.s1 + TH - 1) / TH), (unsigned)N);
  // a workgroup walks its k levels one after the other: with fewer workgroups than CUs (cfg-1: 8 fields x 12 windows) the k
  // small launches finish sooner than one long one
  if ((int64_t)grid.x * grid.y < 256) return ADVCHAIN_ERR_UNSUPPORTED;
  if (query) return ADVCHAIN_OK;
  hipLaunchKernelGGL((k_expo_fused_fwd2d<NT, PPW>), grid, dim3(NT), lds, stream, phi0, fields, F, d, k, TH, halos, disp_rows, fail_flag);
  return ADVCHAIN_OK;
}
