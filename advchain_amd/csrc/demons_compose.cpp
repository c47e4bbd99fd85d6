// Composite entries: ONE call enqueues a whole DemonsCompose direction (adv_morph.py:454-491 for the reference's own call:
// identity initial deformation, smooth = True, the 9-tap window) for the paired field [v; -v] of a solver step.  Nothing
// new runs on the device -- the launches are those of the entry points called below, in the order the Python operator used
// to issue them one ctypes call at a time (round 4: six calls forward, five backward; ~25 us of host time each) --
// but the host leaves the interpreter once per direction.
#include <stdint.h>
#include "advchain_hip.h"

#define ADVCHAIN_OK 0
#define ADVCHAIN_ERR_ARG (-1)
extern "C" void advchain_set_error_(const char* msg);
bool advchain_gauss_xy_takes(int ndim, const int64_t* dims, int64_t planes);      // fields.hip

namespace {
inline int64_t prod(int ndim, const int64_t* d) {
  int64_t p = 1;
  for (int i = 0; i < ndim; ++i) p *= d[i];
  return p;
}
inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}
}  // namespace

extern "C" {

// forward: vel (N, d, g...) -> s1 = [G(s v); G(-s v)] (2N, d, g...) -> phi0 = id + up(s1) / 2^n (2N, d, S...) -> n squarings
// (fields (n-1, 2N, d, S...), pos) -> q = G(border_identity(pos) - id) + id (2N, d, S...) -> rows_max = row maxima of disp.
// 2D only (the 3D chain reads its step count back between the first two launches).  Returns ADVCHAIN_ERR_UNSUPPORTED (-2)
// when one of the fast kernels does not take the shape: nothing has been enqueued then and the caller issues the separate calls.
int advchain_demons_compose_pair_fwd(const float* vel, float* s1, float* phi0, float* fields, float* pos, float* q,
                                     float* disp, float* rows_max, const int32_t* itab, const float* ftab,
                                     const int64_t* S3, const int64_t* g3, const int64_t* B3, int64_t N, int ndim, int n,
                                     const int32_t* hints, const float* weights9_host, float scale, float inv, int fuse,
                                     void* stream) {
  if (!(vel && s1 && phi0 && pos && q && itab && ftab && S3 && g3 && B3 && weights9_host) || ndim != 2 || n < 1 || (n > 1 && !fields)) {
    advchain_set_error_("demons_compose_pair_fwd: null pointer / bad arguments (2D only)");
    return ADVCHAIN_ERR_ARG;
  }
  const int64_t* gd = g3 + 1;      // (2D tables carry a trivial leading axis)
  const int64_t* Sd = S3 + 1;
  // nothing is enqueued unless every launch below takes the shape (the caller then issues the separate calls)
  if (prod(ndim, gd) > 4096 || !advchain_gauss_xy_takes(ndim, Sd, 2 * N * ndim) || !aligned16(pos, q)) return -2;
  // smoothing + upsampling: one launch where the low-resolution plane fits the fused kernel (round 6), else the two calls
  int rc = advchain_tp_interp_fwd_smoothed_pair(vel, s1, phi0, itab, ftab, S3, g3, B3, N * ndim, ndim, 1, inv, disp, weights9_host,
                                                scale, stream);
  if (rc == -2) {
    rc = advchain_gauss_small_pair(vel, s1, N * ndim, ndim, gd, weights9_host, scale, 0, stream);
    if (rc != 0) return rc;
    rc = advchain_tp_interp_fwd(s1, phi0, itab, ftab, S3, g3, B3, 2 * N * ndim, ndim, ndim, 1, inv, nullptr, disp, stream);
  }
  if (rc != 0) return rc;
  // from here on launches are queued: a -2 of an inner entry can no longer mean "nothing was enqueued"  -- the
  // pre-check above is what promises the shapes; if the two ever drift apart the caller must not re-issue the chain on a
  // displacement accumulator that already holds this call's partial state
  auto late = [](int r, const char* what) {
    if (r == -2) { advchain_set_error_(what); return ADVCHAIN_ERR_ARG; }
    return r;
  };
  const int64_t slots = ADVCHAIN_DISP_SLOTS;
  rc = advchain_expo_chain_fwd(phi0, fields, pos, 2 * N, ndim, Sd, n, disp, hints, (disp && fuse) ? disp + (int64_t)(n + 1) * slots : nullptr, stream);
  if (rc != 0) return late(rc, "demons_compose_pair_fwd: the chain refused a shape the pre-check had accepted (launches already queued)");
  rc = advchain_gauss_xy(pos, q, nullptr, 2 * N * ndim, ndim, ndim, Sd, weights9_host, 2, 1, 1.0f, stream, nullptr, 2 * N * ndim);
  if (rc != 0) return late(rc, "demons_compose_pair_fwd: the final smoothing refused a shape the pre-check had accepted (launches already queued)");
  if (disp && rows_max) rc = advchain_slot_rows_max(disp, rows_max, n + 2, slots, 1, stream);
  return rc;
}

// backward: (gq_lo, gq_hi) = the gradients of the two halves of q -> gpos (adjoint of the final smoothing) -> the n adjoint
// squarings -> the upsampling adjoint of (g - gpos) / 2^n (x pass dense, y pass banded) -> gvel = G(s gs1[:N]) - G(s gs1[N:]).
// g, scratch, gpos: field-sized buffers (2N, d, S...); t1: (2N, d, S0, g1); gs1: (2N, d, g...); ws: advchain_scatter_workspace.
int advchain_demons_compose_pair_bwd(const float* gq_lo, const float* gq_hi, const float* pos, const float* phi0,
                                     const float* fields, float* gpos, float* g, float* scratch, int32_t* ws, float* t1,
                                     float* gs1, float* gvel, const int32_t* halos, const int32_t* itab, const float* ftab,
                                     const float* wd, const int32_t* wlo, int64_t WB, const int64_t* S3, const int64_t* g3,
                                     const int64_t* B3, int64_t N, int ndim, int n, const float* weights9_host, float scale,
                                     float inv, void* stream) {
  if (!(gq_lo && gq_hi && pos && phi0 && gpos && g && scratch && ws && t1 && gs1 && gvel && halos && itab && ftab && wd && wlo &&
        weights9_host) || ndim != 2 || n < 1) {
    advchain_set_error_("demons_compose_pair_bwd: null pointer / bad arguments (2D only)");
    return ADVCHAIN_ERR_ARG;
  }
  const int64_t* gd = g3 + 1;
  const int64_t* Sd = S3 + 1;
  const int64_t planes = 2 * N * ndim;
  if (prod(ndim, gd) > 4096 || !advchain_gauss_xy_takes(ndim, Sd, planes) || !aligned16(gq_lo, gq_hi, gpos, pos)) return -2;
  int rc = advchain_gauss_xy(gq_lo, gpos, pos, planes, ndim, ndim, Sd, weights9_host, 0, 2, 1.0f, stream, gq_hi, N * ndim);
  if (rc != 0) return rc;
  rc = advchain_expo_chain_bwd(gpos, phi0, fields, g, scratch, ws, halos, 2 * N, ndim, Sd, n, stream);
  if (rc == -2) {      // (launches are queued: see the forward entry)
    advchain_set_error_("demons_compose_pair_bwd: the chain refused a shape the pre-check had accepted (launches already queued)");
    return ADVCHAIN_ERR_ARG;
  }
  if (rc != 0) return rc;
  // W^T along x (innermost; densified bands where that kernel takes the shape) with the fused (g - gpos) * inv, then along y
  rc = advchain_band_reduce_rows_dense(g, gpos, t1, wd, wlo, planes * S3[1], S3[2], g3[2], WB, inv, stream);
  if (rc == -2) rc = advchain_band_reduce_axis(g, gpos, t1, itab, ftab, S3, g3, B3, 2, planes * S3[1], 1, inv, stream);
  if (rc != 0) return rc;
  rc = advchain_band_reduce_axis(t1, nullptr, gs1, itab, ftab, S3, g3, B3, 1, planes, g3[2], 1.0f, stream);
  if (rc != 0) return rc;
  return advchain_gauss_small_pair(gs1, gvel, N * ndim, ndim, gd, weights9_host, scale, 1, stream);
}

}  // extern "C"
