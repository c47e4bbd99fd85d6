// z-marching gather-form adjoint for 3D warps with an EXACT sub-voxel displacement bound (gfx950).
//
// Same mathematics as adjoint_gather.hip (H = 1):  grad_in[u] = sum_{s in u +- 1} grad_out[s] * prod_a tent(p_s,a - u_a)
// plus the coordinate path at s = u -- but organised around the SAMPLES instead of the outputs, and marched along z:
//
//   * a workgroup owns TY output rows (whole x rows, lane <-> x) and walks ZC planes.  Each sample plane is staged
//     ONCE per workgroup (16-byte loads, offsets o = unnormalize(grid) - s computed at staging time) and each staged
//     sample row is read ONCE per wave that needs it: its clip, its three x tents and its three z tents are computed
//     once and deposited on the 3 (z) x 3 (x) x RPW (y) outputs it touches.  The output-centric kernel re-reads and
//     re-derives every sample row for each of the 9 output rows around it (x2.4 the VALU work, x3 the LDS reads).
//   * a wave keeps 3 planes of partial sums per owned row in registers; the plane that has seen all three of its
//     sample planes is folded over x with two whole-wave DPP shifts and stored, and its register set starts over as the
//     plane two steps ahead (the march loop is unrolled by three: no register moves).
//   * the next plane is requested a step ahead and written to LDS at the end of the step (one barrier per step): the
//     loads of step k+1 are in flight under the arithmetic of step k; where it pays (MarchCfg::LATE_FETCH) the order at
//     the end of a step is commit -> next requests -> stores, so that no wait for loads also waits for the step's own
//     stores.  The image / field planes the coordinate path needs (z-1, z, z+1 around a sample) live in a 4-slot ring,
//     staged two steps ahead.
//   * with |p - s| < 1 guaranteed the tents need no general max(0, 1 - |f - k|): t(-1) = max(0, -f), t(+1) = max(0, f),
//     t(0) = 1 - t(-1) - t(+1); the lower corner of an axis is s + floor(f) with floor(f) in {-1, 0}.
//   * staged rows have 4 zero floats on either side of their 64 voxels (the right zeros of a row ARE the left zeros of the
//     next row in memory: 68 floats a row): a corner at x = -1 or x = S2 reads the 0 that zeros padding asks for,
//     without a select; rows and planes outside the volume are staged as zeros.
//   * results leave through LDS: a wave writes its rows lane <-> x and reads them back 4 voxels per lane, so that two
//     16-byte store instructions replace eight 4-byte ones (a vector-memory instruction costs the CU ~26 clk whether
//     it carries 4 or 16 bytes per lane).
//   * padding mode and clamp are template parameters: run-time selects on per-lane conditions cost scalar-ALU
//     instructions (64-bit lane masks), and the first version of this kernel issued 470 of them per wave and step.
//
// Contract: the caller guarantees |unnormalize(grid) - s| < 1 voxel for every sample (ops.squaring_halo / ops.warp_halo
// measure it in the forward).  Shapes outside the fast form (rows longer than 64 voxels, rows not a multiple of 4,
// misaligned bases) return ADVCHAIN_ERR_UNSUPPORTED and the caller keeps the tile kernel of adjoint_gather.hip.
#include <stdlib.h>
#include <type_traits>
#include "sampler_common.h"

namespace advchain {

// MODE: 0 = zeros padding, positions as given; 1 = positions clipped to [0, S-1] (clamp_grid), zero coordinate gradient
// beyond the border; 2 = border padding (clipped, zero coordinate gradient AT and beyond the border)
enum { kMarchFree = 0, kMarchClamp = 1, kMarchBorder = 2 };
// timing experiments only (ADVCHAIN_MARCH_DEBUG, results are wrong): switch a phase off
constexpr int kDbgNoA = 16, kDbgNoB = 32, kDbgNoStore = 64, kDbgNoStage = 128;
constexpr int kSegOwn = 56;       // owned lanes of an x segment (rows longer than 64 voxels)
constexpr int kMarchXcd = 256;    // workgroup -> tile map that keeps halo-sharing tiles on one XCD (one L2)

// WIDE (rows of 68 .. 80 voxels, cfg-5's 160 x 160 x 80): the x fold through whole-wave DPP shifts ties a lane to an x,
// so a row cannot be cut into flat items as in the forward sampler.  Instead three A waves own RPW rows each, lanes <-> x
// 0 .. 63 (outputs 0 .. 59), and ONE B wave takes the tails of all 3 * RPW rows: three sub-rows of 21 lanes <-> x
// 59 .. 79 (outputs 60 .. 79; the lane at x = 59 is a halo sample whose deposit on x - 1 is dropped, so the DPP shift
// that crosses into the previous sub-row carries a zero).  Four wave-passes per 3 rows of 80 voxels instead of the six
// of two x segments with 56 owned lanes.
constexpr int kWideSplit = 60;    // first output x of the B wave (a multiple of 4: 16-byte stores)
constexpr int kWideSub = 21;      // lanes per sub-row of the B wave: x = 59 .. 79
constexpr int kWidePW = 80;       // longest row of the wide form

template <int C, bool SELF, bool GG, int NW, int RPW, bool WIDE = false>
struct MarchCfg {
  static constexpr int TY = WIDE ? 3 * RPW : NW * RPW;
  static constexpr int R = TY + 2;                              // staged rows per plane (one halo row each side)
  static constexpr int NT = NW * 64;
  // 4 zeros | 64 (80) voxels; the 4 zeros to the RIGHT of a row are the left zeros of the next row in memory (every staged
  // row of every slot is followed by another one; the last one by a 4-float pad): 68 instead of 72 floats per row is what lets
  // three workgroups of the self-composition share a CU's 160 KiB (53 KiB each)
#ifdef ADVCHAIN_MARCH_PAD8      // A/B build: rows with their own right zeros (72 / 88 floats)
  static constexpr int PITCH = WIDE ? kWidePW + 8 : 72;
#else
  static constexpr int PITCH = WIDE ? kWidePW + 4 : 68;
#endif
  static constexpr int QPR = WIDE ? kWidePW / 4 : 16;           // 16-byte staging items per row
  static_assert(!WIDE || NW == 4, "wide form: three A waves and one B wave");
  static constexpr bool HAS_IMG = GG && !SELF;
  static constexpr int RING_CH = SELF ? 3 : (HAS_IMG ? C : 0);  // planes z-1..z+1 are needed: 4-slot ring
  static constexpr int LATE_CH = SELF ? 3 : 3 + C;              // only the current plane is needed: 2 slots
  static constexpr int PS = RING_CH * R * PITCH;                // floats per ring slot
  static constexpr int LS = LATE_CH * R * PITCH;                // floats per late slot
  static constexpr int NA = SELF ? 3 : C + (GG ? 3 : 0);        // arrays written per step (grad_in channels, grad_grid)
  static constexpr int NA_ROUND = NA > 2 ? 2 : NA;              // arrays transposed per round
  static constexpr int TRW = NA_ROUND * RPW * 64;               // transposition scratch per wave (floats)
  static constexpr int ROWS_END = 4 * PS + 2 * LS + 4;          // staged rows, then the right zeros of the last one
  static constexpr bool LATE_FETCH = SELF || (GG && C > 1);     // order of requests / commit / stores at the end of a step (see there)
  static constexpr size_t LDS = (size_t)(ROWS_END + NW * TRW) * sizeof(float);
  // per SIMD: 128 / 168 / 256 VGPRs (512-thread blocks: 2 waves per SIMD each).  Without the SLP vectoriser (build.py) the
  // self-composition needs 152: three workgroups a CU
  static constexpr int MIN_WAVES = (C == 1) ? 4 : (SELF ? (RPW == 1 && NW == 8 ? 4 : 3) : 2);
  static_assert(R * QPR <= NT, "one staging item (4 voxels of one row, all channels) per thread");
  static_assert(!SELF || C == 3, "the self-composition carries 3 channels");
};

__device__ __forceinline__ float march_unnormalize(float g, int S) { return ((g + 1.f) * 0.5f) * (float)(S - 1); }

template <int C, bool SELF, bool GG, int MODE, int NW, int RPW, bool WIDE = false>
__global__ void __launch_bounds__(NW * 64, (MarchCfg<C, SELF, GG, NW, RPW, WIDE>::MIN_WAVES))
k_adjoint_march(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int zc, int flags,
                int32_t* __restrict__ untracked, int nseg) {
  using G = MarchCfg<C, SELF, GG, NW, RPW, WIDE>;
  constexpr int R = G::R, TY = G::TY, RC = G::RING_CH, LC = G::LATE_CH, P = G::PITCH, PS = G::PS, LS = G::LS;
  constexpr bool CLIP = MODE != kMarchFree, BORDER = MODE == kMarchBorder;
  if (untracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) untracked[3] = -1;   // see adjoint_gather.hip
  extern __shared__ float lds[];
  float* const ring = lds;                       // [slot 4][RC][R][P]
  float* const late = lds + 4 * PS;              // [slot 2][LC][R][P]
  float* const trbuf = lds + G::ROWS_END;        // [wave][TRW]
  const int V = (int)d.voxels();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup L runs on XCD L % 8 (observed dispatch order; speed only): give every XCD a contiguous run of tiles, so
  // that the halo rows / planes two neighbouring workgroups both stage are served by the same L2
  int tile = blockIdx.x + gridDim.x * blockIdx.y;
  const int tiles = gridDim.x * gridDim.y;
  if ((flags & kMarchXcd) && (tiles & 7) == 0) tile = (tile & 7) * (tiles >> 3) + (tile >> 3);
  const int n = tile / (int)gridDim.x;
  int rem = tile - n * (int)gridDim.x;
  // rows longer than 64 voxels: x segments of 56 owned lanes with 4 halo lanes either side (16-byte aligned staging)
  const int seg = rem % nseg;
  rem /= nseg;
  const int xbase = nseg > 1 ? seg * kSegOwn - 4 : 0;      // x of lane 0
  const int ty = rem % n1, tz = rem / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* gn = grid + (int64_t)n * 3 * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
  float* ggn = GG ? ggrid + (int64_t)n * 3 * V : nullptr;
  const int S[3] = {d.s2, d.s1, d.s0};
  const int plane_stride = d.s1 * d.s2;

  // ---- the zero columns of every staged row (never written again)
  for (int row = threadIdx.x; row <= (4 * RC + 2 * LC) * R; row += G::NT)      // (the last one: the pad behind the rows)
    *reinterpret_cast<float4*>(lds + row * P) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- staging item of this thread: 4 consecutive x of staged row r_st, every channel
  const bool has_item = threadIdx.x < R * G::QPR;
  const int r_st = (int)threadIdx.x / G::QPR, q_st = (int)threadIdx.x % G::QPR;
  const int sy_st = y0 - 1 + r_st, x_st = xbase + 4 * q_st;
  const bool row_ok = has_item && sy_st >= 0 && sy_st < d.s1 && x_st >= 0 && x_st < d.s2;
  const int lds_item = r_st * P + 4 + 4 * q_st;

  // Loads are UNCONDITIONAL, from addresses clamped into the volume, and what lies outside is zeroed with selects when
  // the values go to LDS: a load inside `if (inside)` gets its own exec-mask block, and the compiler put a full
  // `s_waitcnt vmcnt(0)` between two such blocks -- one exposed memory round trip per step.
  const int sy_c = min(max(sy_st, 0), d.s1 - 1), x_c = (x_st >= 0 && x_st < d.s2) ? x_st : 0;
  const int row_off = sy_c * d.s2 + x_c;
  auto plane_ok = [&](int p) { return row_ok && p >= 0 && p < d.s0; };
  auto load_rows = [&](const float* base, int nch, int p, float (*v)[4]) {
    const uint32_t s = (uint32_t)(min(max(p, 0), d.s0 - 1) * plane_stride + row_off);
#pragma unroll 4
    for (int c = 0; c < nch; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(base + (size_t)c * V + s);
      v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
    }
  };
  auto zero_outside = [&](int p, int nch, float (*v)[4]) {
    const bool ok = plane_ok(p);
#pragma unroll 4
    for (int c = 0; c < nch; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[c][k] = ok ? v[c][k] : 0.f;
  };
  // field plane p -> offsets o = unnormalize(field) - own voxel (what the tents and the corner search work on)
  auto field_to_offsets = [&](int p, float (*v)[4]) {
    const bool ok = plane_ok(p);          // rows / planes outside the volume: offset 0 (their grad_out is 0 as well)
    const float sa[3] = {(float)x_st, (float)sy_st, (float)p};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xs = march_unnormalize(v[a][k], S[a]);
        v[a][k] = ok ? xs - (a == 0 ? sa[0] + (float)k : sa[a]) : 0.f;
      }
  };
  auto store_lds = [&](float* slot_base, int ch, const float (&v)[4]) {
    *reinterpret_cast<float4*>(slot_base + ch * R * P + lds_item) = make_float4(v[0], v[1], v[2], v[3]);
  };
  // ring plane p: SELF -> field offsets; warp with grad_grid -> the image.  late plane p: SELF -> grad_out; warp ->
  // field offsets + grad_out
  constexpr int RCA = RC > 0 ? RC : 1;
  auto fetch_ring = [&](int p, float (*v)[4]) {
    if constexpr (SELF) load_rows(gn, 3, p, v);
    else if constexpr (G::HAS_IMG) load_rows(inn, C, p, v);
  };
  auto commit_ring = [&](int p, float (*v)[4]) {
    if constexpr (RC > 0) {
      if constexpr (SELF) field_to_offsets(p, v);
      else zero_outside(p, RC, v);
#pragma unroll
      for (int c = 0; c < RC; ++c) store_lds(ring + (p & 3) * PS, c, v[c]);
    }
  };
  auto fetch_late = [&](int p, float (*v)[4]) {
    if constexpr (SELF) load_rows(gon, 3, p, v);
    else {
      load_rows(gn, 3, p, v);
      load_rows(gon, C, p, v + 3);
    }
  };
  auto commit_late = [&](int p, float (*v)[4]) {
    if constexpr (!SELF) field_to_offsets(p, v);
    zero_outside(p, SELF ? 3 : C, SELF ? v : v + 3);
#pragma unroll
    for (int c = 0; c < LC; ++c) store_lds(late + (p & 1) * LS, c, v[c]);
  };

  // ---- prologue: ring planes za-1, za; late plane za-1 -- every load is issued before the first LDS write (one
  // memory round trip, not three)
  float pr[RCA][4], pl[LC][4];
  if (has_item) {
    float pr0[RCA][4];
    fetch_ring(za - 1, pr0);
    fetch_ring(za, pr);
    fetch_late(za - 1, pl);
    commit_ring(za - 1, pr0);
    commit_ring(za, pr);
    commit_late(za - 1, pl);
    if constexpr (G::LATE_FETCH) {     // the planes the FIRST step commits at its end: requested now (see the end of a step)
      fetch_ring(za + 1, pr);
      fetch_late(za, pl);
    }
  }
  __syncthreads();

  // partial sums: [owned row][target plane zp-1, zp, zp+1][channel][deposit on x-1, x, x+1]
  float acc[RPW][3][C][3];
#pragma unroll
  for (int o = 0; o < RPW; ++o)
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[o][kz][c][kx] = 0.f;
  float gg_hold[RPW][3];     // SELF: coordinate-path gradient of the previous plane's samples (added when that plane is stored)
#pragma unroll
  for (int o = 0; o < RPW; ++o) gg_hold[o][0] = gg_hold[o][1] = gg_hold[o][2] = 0.f;

  // x and first owned output row (within the tile) of this lane.  Narrow rows / x segments / A waves: lane <-> x, the
  // wave's own rows.  B wave of the wide form: sub-row sb of 21 lanes <-> x = 59 .. 79 of rows sb * RPW ..; lane 63 idles
  // on the zero column at x = 80.
  const bool bwave = WIDE && wave == 3;
  const int sb = bwave ? min(lane / kWideSub, 2) : 0;
  const int xl = bwave ? kWideSplit - 1 + lane - kWideSub * sb : xbase + lane;
  const int own_row0 = bwave ? sb * RPW : wave * RPW;
  const bool kill_left = bwave && xl == kWideSplit - 1;      // its deposit on x - 1 belongs to an A wave
  const int lbase = own_row0 * P + 4 + (xl - xbase);         // this lane in a staged channel plane: + (ch * R + row) * P
  // where the lane puts a result in the wave's transposition scratch: the 20 owned voxels of sub-row sb at 20 * sb
  const int tr_pos = bwave ? ((xl >= kWideSplit && xl < kWidePW) ? 20 * sb + xl - kWideSplit : 60 + (lane == 63 ? 3 : sb)) : lane;
  const float xlo = -(float)xl, xhi = (float)(d.s2 - 1 - xl);
  const float half_top[3] = {0.5f * (float)(d.s2 - 1), 0.5f * (float)(d.s1 - 1), 0.5f * (float)(d.s0 - 1)};
  float* const tr = trbuf + wave * G::TRW;

  // results of one step: `vals[a][o]` (array a, owned row o) go out 4 voxels per lane through the wave's LDS scratch.
  // Array a of the step starts at dst[a] + row_base[a] (floats), row o at + o * S2; `ok[a]` says whether the array is
  // produced in this step (wave-uniform).
  auto store_rows = [&](float (&vals)[G::NA][RPW], float* const (&dst)[G::NA], const int (&row_base)[G::NA],
                        const bool (&ok)[G::NA]) {
    constexpr int NR = G::NA_ROUND;
#pragma unroll
    for (int a0 = 0; a0 < G::NA; a0 += NR) {
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int o = 0; o < RPW; ++o)
          if (a0 + a < G::NA) tr[(a * RPW + o) * 64 + tr_pos] = vals[a0 + a][o];
      lds_order();
      // items of this round: (array, row, quad); 16 quads per row
      constexpr int ITEMS = NR * RPW * 16;
#pragma unroll
      for (int i0 = 0; i0 < ITEMS; i0 += 64) {
        const int j = i0 + lane;
        const int a = j / (RPW * 16), o = (j / 16) % RPW, q = j & 15;
        const float4 v4 = *reinterpret_cast<const float4*>(tr + (a * RPW + o) * 64 + 4 * q);
        // the quad's x and owned row: lane <-> x (x segments: quads 1 .. 14 are owned) | A wave: outputs 0 .. 59 | B wave:
        // quad q is quad q % 5 of sub-row q / 5
        int xq = xbase + 4 * q, rq = wave * RPW + o;
        bool inside = j < ITEMS && (nseg == 1 || (q >= 1 && q <= 14));
        if (WIDE) {
          inside = j < ITEMS && q <= 14;
          if (wave == 3) { xq = kWideSplit + 4 * (q % 5); rq = (q / 5) * RPW + o; }
        }
        inside = inside && xq < d.s2 && (y0 + rq) < d.s1;
        bool valid = false;
        float* p = nullptr;
#pragma unroll
        for (int aa = 0; aa < NR; ++aa)
          if (a0 + aa < G::NA && a == aa) {
            valid = inside && ok[a0 + aa];
            p = dst[a0 + aa] + (uint32_t)(row_base[a0 + aa] + rq * d.s2 + xq);
          }
        if (valid) *reinterpret_cast<float4*>(p) = v4;
      }
      lds_order();
    }
  };

  // One marching step.  The three planes of partial sums ROTATE through their register sets instead of shifting down one slot
  // per step (the shift was 36 moves + 18 zero-fills per step, copied through temporaries by the register allocator: 125
  // v_mov of the 858 VALU instructions of a step): plane p lives in slot (p - (za - 1)) mod 3, and the loop below runs
  // three copies of the step whose slot indices are compile-time constants (ROT = (zp - (za - 1)) mod 3).
  auto step = [&](const int zp, auto rot_tag) {
    constexpr int ROT = decltype(rot_tag)::value;
    constexpr int SL[3] = {(ROT + 2) % 3, ROT, (ROT + 1) % 3};      // slots of the target planes zp-1, zp, zp+1
    // (LATE_FETCH: the planes this step commits at its end -- ring zp+2, late zp+1 -- were requested at the END of the
    // previous step: see below)
    if constexpr (!G::LATE_FETCH) {
      if (has_item) {
        fetch_ring(zp + 2, pr);
        fetch_late(zp + 1, pl);
      }
    }

    const float* lslot = late + (zp & 1) * LS;
    // (rows below are relative to the lane's first owned row: lbase carries own_row0)
    const float* fbase = (SELF ? ring + (zp & 3) * PS : lslot) + lbase;        // field offsets, 3 ch; [ch * R * P + row * P]
    const float* gobase = (SELF ? lslot : lslot + 3 * R * P) + lbase;          // grad_out, C ch
    const float fzp = (float)zp;
    const float zlo = -fzp, zhi = (float)(d.s0 - 1) - fzp;

    float vals[G::NA][RPW];
    float* dst[G::NA];
    int row_base[G::NA];
    bool ok[G::NA];
    const int zt = zp - 1;
    const bool fin = zt >= za && zt < zb;
    {
      // ---- phase B: deposits of sample plane zp on the target planes zp-1, zp, zp+1
      if (zp >= 0 && zp < d.s0 && !(flags & kDbgNoB)) {
#pragma unroll
        for (int i = 0; i < RPW + 2; ++i) {
          const int r = i;                        // staged row of the sample past the lane's own_row0; sample y = y0 - 1 + own_row0 + r
          float fx = fbase[(0 * R + r) * P];
          float fy = fbase[(1 * R + r) * P];
          float fz = fbase[(2 * R + r) * P];
          float go[C];
#pragma unroll
          for (int c = 0; c < C; ++c) go[c] = gobase[(c * R + r) * P];
          if (CLIP) {
            const float ys = (float)(y0 - 1 + own_row0 + r);
            fx = __builtin_amdgcn_fmed3f(fx, xlo, xhi);
            fy = __builtin_amdgcn_fmed3f(fy, -ys, (float)(d.s1 - 1) - ys);
            fz = __builtin_amdgcn_fmed3f(fz, zlo, zhi);
          }
          float tx[3], tyv[3], tzv[3];
          tx[0] = fmaxf(0.f, -fx); tx[2] = fmaxf(0.f, fx); tx[1] = (1.f - tx[0]) - tx[2];
          if (WIDE) tx[0] = kill_left ? 0.f : tx[0];
          tyv[0] = fmaxf(0.f, -fy); tyv[2] = fmaxf(0.f, fy); tyv[1] = (1.f - tyv[0]) - tyv[2];
          tzv[0] = fmaxf(0.f, -fz); tzv[2] = fmaxf(0.f, fz); tzv[1] = (1.f - tzv[0]) - tzv[2];
#pragma unroll
          for (int o = 0; o < RPW; ++o) {
            const int k = o + 1 - i;              // output row minus sample row (compile time)
            if (k < -1 || k > 1) continue;
            const float wy = tyv[k + 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const float a = go[c] * wy;
#pragma unroll
              for (int kz = 0; kz < 3; ++kz) {
                const float b = a * tzv[kz];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[o][SL[kz]][c][kx] = fmaf(b, tx[kx], acc[o][SL[kz]][c][kx]);
              }
            }
          }
        }
      }
      // ---- target plane zp-1 has now seen its three sample planes: fold over x; its slot starts over as plane zp+2
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int a = (SELF || !GG) ? c : 3 + c;
        dst[a] = ginn + (size_t)c * V;
        row_base[a] = (zt * d.s1 + y0) * d.s2;
        ok[a] = fin;
#pragma unroll
        for (int o = 0; o < RPW; ++o) {
          float v = lane_prev_f(acc[o][SL[0]][c][2]) + acc[o][SL[0]][c][1] + lane_next_f(acc[o][SL[0]][c][0]);
          if (SELF) v += gg_hold[o][c < 3 ? c : 0];
          vals[a][o] = v;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[o][SL[0]][c][kx] = 0.f;
        }
      }
    }

    // ---- phase A: coordinate-path gradient of the owned samples of plane zp (corner values from the ring)
    const bool do_a = (SELF || GG) && zp >= za && zp < zb && !(flags & kDbgNoA);
    if constexpr (!SELF && GG) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dst[a] = ggn + (size_t)a * V;
        row_base[a] = (zp * d.s1 + y0) * d.s2;
        ok[a] = do_a;
#pragma unroll
        for (int o = 0; o < RPW; ++o) vals[a][o] = 0.f;
      }
    }
    if (do_a) {
#pragma unroll
      for (int o = 0; o < RPW; ++o) {
        const int r = o + 1;                    // (past the lane's own_row0, as above)
        const float ys = (float)(y0 - 1 + own_row0 + r);
        const float lo[3] = {xlo, -ys, zlo};
        const float hi[3] = {xhi, (float)(d.s1 - 1) - ys, zhi};
        float w1[3], mult[3];
        int fl[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float f = fbase[(a * R + r) * P];
          float fc = f;
          mult[a] = half_top[a];
          if (CLIP) {
            fc = __builtin_amdgcn_fmed3f(f, lo[a], hi[a]);
            if (BORDER) mult[a] = (f > lo[a] && f < hi[a]) ? mult[a] : 0.f;
            else mult[a] = (fc == f) ? mult[a] : 0.f;
          }
          const float flo = floorf(fc);         // -1 or 0
          w1[a] = fc - flo;
          fl[a] = (int)flo;
        }
        float go[C];
#pragma unroll
        for (int c = 0; c < C; ++c) go[c] = gobase[(c * R + r) * P];
        // corner (0,0,0) of this sample in the ring: plane zp + fl_z, staged row r + fl_y, x = lane + fl_x (the zero
        // columns, rows and planes make a corner outside the volume read 0)
        const float* q0 = ring + ((zp + fl[2]) & 3) * PS + (r + fl[1]) * P + lbase + fl[0];
        const float* q1 = ring + ((zp + fl[2] + 1) & 3) * PS + (r + fl[1]) * P + lbase + fl[0];
        const float wx1 = w1[0], wy1 = w1[1], wz1 = w1[2];
        float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v[2][2][2];
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              const float* p = (cz ? q1 : q0) + (c * R + cy) * P;
              v[cz][cy][0] = p[0];
              v[cz][cy][1] = p[1];
            }
          // SELF: phi_c(u) = (o_c(u) + u_c) * 2/(S_c-1) - 1: only differences along an axis enter; the identity part of a
          // difference along axis c is the index step (1)
          const float ux = (SELF && c == 0) ? 1.f : 0.f, uyy = (SELF && c == 1) ? 1.f : 0.f, uzz = (SELF && c == 2) ? 1.f : 0.f;
          // d/dx: x differences, interpolated over y then z; likewise the other two
          const float ex00 = v[0][0][1] - v[0][0][0], ex01 = v[0][1][1] - v[0][1][0];
          const float ex10 = v[1][0][1] - v[1][0][0], ex11 = v[1][1][1] - v[1][1][0];
          const float ax0 = fmaf(wy1, ex01 - ex00, ex00), ax1 = fmaf(wy1, ex11 - ex10, ex10);
          const float dx = fmaf(wz1, ax1 - ax0, ax0) + ux;
          const float ey00 = v[0][1][0] - v[0][0][0], ey01 = v[0][1][1] - v[0][0][1];
          const float ey10 = v[1][1][0] - v[1][0][0], ey11 = v[1][1][1] - v[1][0][1];
          const float ay0 = fmaf(wx1, ey01 - ey00, ey00), ay1 = fmaf(wx1, ey11 - ey10, ey10);
          const float dy = fmaf(wz1, ay1 - ay0, ay0) + uyy;
          const float ez00 = v[1][0][0] - v[0][0][0], ez01 = v[1][0][1] - v[0][0][1];
          const float ez10 = v[1][1][0] - v[0][1][0], ez11 = v[1][1][1] - v[0][1][1];
          const float az0 = fmaf(wx1, ez01 - ez00, ez00), az1 = fmaf(wx1, ez11 - ez10, ez10);
          const float dz = fmaf(wy1, az1 - az0, az0) + uzz;
          const float kc = SELF ? go[c] * (1.f / half_top[c < 3 ? c : 0]) : go[c];
          acc3[0] = fmaf(dx, kc, acc3[0]); acc3[1] = fmaf(dy, kc, acc3[1]); acc3[2] = fmaf(dz, kc, acc3[2]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if constexpr (SELF) gg_hold[o][a] = mult[a] * acc3[a];
          else vals[a][o] = mult[a] * acc3[a];
        }
      }
    }
    // ---- the prefetched planes go to LDS; nobody reads these slots in this step (ring: zp+2 = zp-2 mod 4, last read
    // in step zp-1; late: zp+1 = zp-1 mod 2, last read in step zp-1).  Order of the memory traffic at the end of a step:
    // commit (wait for the planes requested a step ago), request the next planes, THEN this step's stores.  vmcnt counts in
    // order and the stores sit behind (wave-uniform and per-lane) branches, so a wait for loads that are older than such
    // stores has to be vmcnt(0): with the stores issued just before the commit every step waited for their acknowledgements
    // (measured with the stores switched off: 21 of 119 us at 8 x 3 x 128 x 128 x 64).  Now everything a commit waits for
    // was issued a whole step earlier.  (Stores before the requests does not work either: the register allocator reuses
    // the stores' data registers as load destinations and guards them with the same wait.)
    // (Committed unconditionally, also the planes beyond the chunk that nobody will read -- their slots hold dead planes:
    // a request that is never waited for leaves the compiler guarding its registers at the next request, with a count that
    // includes the stores.)
    // Commit and request share ONE block: a request whose wait does not dominate the next write of its registers leaves
    // the compiler guarding them there, with a count that includes the stores.
    // Same-box A/B per kernel: the self-composition -7..-12 % (109 -> 97-101 us at 8 x 3 x 128 x 128 x 64), C = 4 with
    // grad_grid -3..-7 %; the warps without grad_grid (no phase A: short steps) +16-21 % on rows of 64 (84.6 -> 98.2 us) and
    // the one-channel warp +3-5 % (31.9 -> 33.5 us): those keep requests at the top and stores before the commit.
    if constexpr (G::LATE_FETCH) {
      if (has_item) {
        commit_ring(zp + 2, pr);
        commit_late(zp + 1, pl);
        // (a plane beyond the chunk is loaded from a clamped address: no branch around the loads)
        fetch_ring(zp + 3, pr);
        fetch_late(zp + 2, pl);
      }
      if (!(flags & kDbgNoStore)) store_rows(vals, dst, row_base, ok);
    } else {
      if (!(flags & kDbgNoStore)) store_rows(vals, dst, row_base, ok);
      if (has_item) {
        if (zp + 2 <= zb) commit_ring(zp + 2, pr);
        if (zp + 1 <= zb) commit_late(zp + 1, pl);
      }
    }
    __syncthreads();
  };
  for (int zp = za - 1; zp <= zb; zp += 3) {
    step(zp, std::integral_constant<int, 0>{});
    if (zp + 1 > zb) break;
    step(zp + 1, std::integral_constant<int, 1>{});
    if (zp + 2 > zb) break;
    step(zp + 2, std::integral_constant<int, 2>{});
  }
}

}  // namespace advchain

using namespace advchain;

static int march_zc(const Dims& d, int64_t N, int ty) {
  static const int forced = getenv("ADVCHAIN_MARCH_ZC") ? atoi(getenv("ADVCHAIN_MARCH_ZC")) : 0;   // A/B knob (0: the rule below)
  if (forced > 0) return forced;
  // enough workgroups to fill 256 CUs four deep, but chunks no shorter than 8 planes (2 of ZC + 2 steps are halo work)
  const int64_t cols = N * ((d.s1 + ty - 1) / ty);
  int zc = d.s0;
  while (zc > 8 && cols * ((d.s0 + zc - 1) / zc) < 1024) zc = (zc + 1) / 2;
  return zc;
}

template <int C, bool SELF, bool GG, int MODE, int NW, int RPW, bool WIDE = false>
static void launch_march(const float* gout, const float* in, const float* grid, float* gin, float* ggrid, int64_t N,
                         Dims d, int32_t* ws, hipStream_t st) {
  using G = MarchCfg<C, SELF, GG, NW, RPW, WIDE>;
  auto kern = k_adjoint_march<C, SELF, GG, MODE, NW, RPW, WIDE>;
  // occupancy experiment (round 6, profiles/r06/f1_3d/): see launch_fwd_march (sample_march.hip)
  static const size_t lds_pad = getenv("ADVCHAIN_MARCH_LDS_PAD") ? (size_t)atoi(getenv("ADVCHAIN_MARCH_LDS_PAD")) * 1024 : 0;
  static bool attr_set = false;
  if (G::LDS + lds_pad > 65536 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::LDS + lds_pad));
    attr_set = true;
  }
  static const int dbg = (getenv("ADVCHAIN_MARCH_DEBUG") ? atoi(getenv("ADVCHAIN_MARCH_DEBUG")) : 0) |
                         (kMarchXcd);
  const int n1 = (d.s1 + G::TY - 1) / G::TY;
  const int nseg = (WIDE || d.s2 <= 64) ? 1 : (d.s2 + kSegOwn - 1) / kSegOwn;
  const int zc = march_zc(d, N * nseg, G::TY);
  const int n0 = (d.s0 + zc - 1) / zc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nseg * n1 * n0), (unsigned)N), dim3(G::NT), G::LDS + lds_pad, st, gout, in, grid, gin, ggrid, d,
                     n1, zc, dbg, SELF ? ws : (int32_t*)nullptr, nseg);
}

// rows of 68 .. 80 voxels: the A / B wave form (MarchCfg, WIDE)
static bool march_wide(const Dims& d) {
  static const bool off = getenv("ADVCHAIN_NO_WIDE_ADJOINT") != nullptr;   // A/B knob
  return !off && d.s2 > 64 && d.s2 <= kWidePW;
}

static bool march_shape_ok(const Dims& d, const void* a, const void* b, const void* c, const void* e, const void* f) {
  static const bool off = getenv("ADVCHAIN_NO_MARCH_ADJOINT") != nullptr;   // A/B knob
  if (off) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                       reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f);
  return d.s2 >= 8 && (d.s2 & 3) == 0 && (al & 15) == 0 && d.s0 >= 2 && d.voxels() * 4 < (1ll << 31);
}

// Exact-bound (|displacement| < 1 voxel) self-composition backward, 3D.  ADVCHAIN_ERR_UNSUPPORTED: use the tile kernel.
int advchain_self_adjoint_march_launch(const float* gout, const float* phi, float* gphi, int64_t N, Dims d,
                                       int32_t* workspace, hipStream_t st) {
  if (!march_shape_ok(d, gout, phi, gphi, nullptr, nullptr)) return ADVCHAIN_ERR_UNSUPPORTED;
  static const int rpw = 2;   // measured optimum (was a tuning knob until round 4)
  if (march_wide(d)) launch_march<3, true, false, kMarchBorder, 4, 2, true>(gout, phi, phi, gphi, nullptr, N, d, workspace, st);
  else if (rpw == 2) launch_march<3, true, false, kMarchBorder, 4, 2>(gout, phi, phi, gphi, nullptr, N, d, workspace, st);
  else if (rpw == 8) launch_march<3, true, false, kMarchBorder, 8, 1>(gout, phi, phi, gphi, nullptr, N, d, workspace, st);
  else launch_march<3, true, false, kMarchBorder, 4, 1>(gout, phi, phi, gphi, nullptr, N, d, workspace, st);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

template <int MODE>
static void launch_warp_march(const float* gout, const float* in, const float* grid, float* gin, float* ggrid, int64_t N,
                              int64_t C, Dims d, hipStream_t st) {
  if (C == 1 && march_wide(d)) {
    if (ggrid) launch_march<1, false, true, MODE, 4, 2, true>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
    else launch_march<1, false, false, MODE, 4, 2, true>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
  } else if (C == 1) {
    if (ggrid) launch_march<1, false, true, MODE, 4, 2>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
    else launch_march<1, false, false, MODE, 4, 2>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
  } else {
    if (ggrid) launch_march<4, false, true, MODE, 4, 1>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
    else launch_march<4, false, false, MODE, 4, 1>(gout, in, grid, gin, ggrid, N, d, nullptr, st);
  }
}

// Exact-bound grid_sample backward (grad_in [+ grad_grid]), 3D, C in {1, 4}, zeros / border padding.
int advchain_warp_adjoint_march_launch(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                       int64_t N, int64_t C, Dims d, int padding, int clamp_grid, hipStream_t st) {
  if (padding == PAD_REFLECTION || (C != 1 && C != 4) || !gin) return ADVCHAIN_ERR_UNSUPPORTED;
  if (!march_shape_ok(d, gout, grid, gin, ggrid, ggrid ? in : nullptr)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (padding == PAD_BORDER) launch_warp_march<kMarchBorder>(gout, in, grid, gin, ggrid, N, C, d, st);
  else if (clamp_grid) launch_warp_march<kMarchClamp>(gout, in, grid, gin, ggrid, N, C, d, st);
  else launch_warp_march<kMarchFree>(gout, in, grid, gin, ggrid, N, C, d, st);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
