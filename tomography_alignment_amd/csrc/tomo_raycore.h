// tomo_raycore.h -- per-projection constants and per-ray sample arithmetic shared by every kernel.
//
// Semantics restated (not translated) from the reference:
//   pose / ray set-up   utilities/ray_voxel_utilities.py:6-12,72-94   (float64, as the reference)
//   floor / weights     utilities/ray_voxel_utilities.py:96-99
//   corner rule         src/ray_wt_grad.f90:35-89 (each corner contributes iff in bounds)
//   pose Jacobian       utilities/ray_voxel_utilities.py:15-50
//
// Design (MI355X-first): a parallel-beam projection is an affine lattice
//     p(ix, iz, j) = p0 + ix*u + iz*w + j*d            (index space = world - vox_origin)
// so a projection is 13 doubles of constants instead of the reference's (3, n_rays, n) float64
// temporaries.  Volumes are read through a zero halo of TOMO_HALO voxels, which turns the
// per-corner bounds tests into plain loads of zeros (identical result: an out-of-bounds corner
// contributes 0).  Sample positions inside a block of TOMO_JB samples are float32 offsets from a
// float64 integer anchor at the block's middle, so coordinates keep ~1e-6 voxel accuracy at 1024^3 where
// plain float32 (ulp 6e-5 at 1024) would not.
#ifndef TOMO_RAYCORE_H_
#define TOMO_RAYCORE_H_
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TOMO_HD __host__ __device__ __forceinline__
#else
#define TOMO_HD inline
#endif

#define TOMO_HALO 2   // zero voxels on every side of the padded volume
#define TOMO_JB 32    // samples per re-anchored block in the ray-driven kernels (in-block coordinates within +-17 of the anchor)

struct TomoGeomC {    // device-side copy of tomo_geom
    int32_t nx, ny, nz, ndx, ndz;
    int32_t nxp, nyp, nzp;   // padded dims = n + 2*TOMO_HALO
    double org[3], det_x0, det_z0, det_dx, det_dz, src_y, det_y, step;
};

struct ProjC {        // one projection's ray lattice (index space)
    double p0[3], u[3], w[3], d[3];
    double rlen;      // |r_0|                       ray_voxel_utilities.py:86
    int32_t n;        // samples per ray = int(rlen/step)   :88
    int32_t pad_;
};

struct GradC {        // extras for the 6-DoF pose Jacobian of one projection (:25-49)
    double rzx[3][3];                   // Rz*Rx            (der rows 0-2 are its columns)
    double a3[3][3], a4[3][3], a5[3][3]; // dRz*Rx, Rz*dRx, Rz*Rx*dRy
    double ry[3][3], t[3];
    double s00[3], sdx, sdz;            // untransformed (cor-shifted) source of ray (0,0) and pitches
    double app[3][3];                   // der rows 6-8: the three operators applied to (d - s)
    int32_t b_row;                      // fused cost/gradient: row of the measured-projection table this pose is compared with
    int32_t slot;                       // ... and the caller's index of this pose (where its 7 sums / residual row go)
};

struct TomoM3 { double m[3][3]; };
static inline TomoM3 tomo_rz(double a) { double c = cos(a), s = sin(a); return {{{c, -s, 0}, {s, c, 0}, {0, 0, 1}}}; }
static inline TomoM3 tomo_rx(double a) { double c = cos(a), s = sin(a); return {{{1, 0, 0}, {0, c, -s}, {0, s, c}}}; }
static inline TomoM3 tomo_ry(double a) { double c = cos(a), s = sin(a); return {{{c, 0, s}, {0, 1, 0}, {-s, 0, c}}}; }
static inline TomoM3 tomo_drz(double a) { double c = cos(a), s = sin(a); return {{{-s, -c, 0}, {c, -s, 0}, {0, 0, 0}}}; }
static inline TomoM3 tomo_drx(double a) { double c = cos(a), s = sin(a); return {{{0, 0, 0}, {0, -s, -c}, {0, c, -s}}}; }
static inline TomoM3 tomo_dry(double a) { double c = cos(a), s = sin(a); return {{{-s, 0, c}, {0, 0, 0}, {-c, 0, -s}}}; }
// Inner products of length 3 are rounded the way numpy's np.dot rounds them in the reference (BLAS dgemm on an FMA machine: one
// rounded product, then two fused multiply-adds in ascending k).  This matters for exactly one number: |r_0| is an integer up to
// rounding, n = int(|r_0| / step) (ray_voxel_utilities.py:88) is K or K - 1 by that rounding, and for volumes longer in x than in y
// the last sample lies inside the object.  With plain a*b + c*d + e*f the library's n differed from numpy's for 8 % of random poses;
// with this form |r_0| agrees bit for bit (3000 of 3000 poses, tools/n_check.py).
static inline double tomo_dot3(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
static inline TomoM3 tomo_mm(const TomoM3 &a, const TomoM3 &b)
{
    TomoM3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = tomo_dot3(a.m[i][0], b.m[0][j], a.m[i][1], b.m[1][j], a.m[i][2], b.m[2][j]);
    return r;
}
static inline void tomo_mv(const TomoM3 &a, const double x[3], double o[3])
{
    for (int i = 0; i < 3; ++i) o[i] = tomo_dot3(a.m[i][0], x[0], a.m[i][1], x[1], a.m[i][2], x[2]);
}

// pose = phi, alpha, beta, tx, ty, tz, cor_x
static inline void tomo_make_projc(const TomoGeomC &g, const double *pose, ProjC &c, GradC *gc)
{
    const double phi = pose[0], alpha = pose[1], beta = pose[2];
    const double t[3] = {pose[3], pose[4], pose[5]};
    const double cor = pose[6];
    TomoM3 Rz = tomo_rz(phi), Rx = tomo_rx(alpha), Ry = tomo_ry(beta);
    TomoM3 Rzx = tomo_mm(Rz, Rx);                                   // :8
    TomoM3 M = tomo_mm(Rzx, Ry);
    const double s00[3] = {g.det_x0 + cor, g.src_y, g.det_z0};      // :72  (geometry.py:99)
    const double d00[3] = {g.det_x0 + cor, g.det_y, g.det_z0};      // :73  (geometry.py:100)
    double q[3], ps[3], pd[3];
    tomo_mv(Ry, s00, q);
    for (int a = 0; a < 3; ++a) q[a] += t[a];                       // :9
    tomo_mv(Rzx, q, ps);                                            // :10
    tomo_mv(Ry, d00, q);
    for (int a = 0; a < 3; ++a) q[a] += t[a];
    tomo_mv(Rzx, q, pd);
    double r[3];
    volatile double sq[3];                                          // np.linalg.norm: three rounded squares, added in order (no contraction)
    for (int a = 0; a < 3; ++a) {
        c.p0[a] = ps[a] - g.org[a];                                 // :74
        r[a] = (pd[a] - g.org[a]) - c.p0[a];                        // :85
        sq[a] = r[a] * r[a];
        c.u[a] = M.m[a][0] * g.det_dx;
        c.w[a] = M.m[a][2] * g.det_dz;
    }
    c.rlen = sqrt((sq[0] + sq[1]) + sq[2]);                         // :86
    for (int a = 0; a < 3; ++a) c.d[a] = g.step * (r[a] / c.rlen);  // :87,93
    c.n = (int32_t)(c.rlen / g.step);                               // :88
    c.pad_ = 0;
    if (gc) {
        TomoM3 dRz = tomo_drz(phi), dRx = tomo_drx(alpha), dRy = tomo_dry(beta);
        TomoM3 A3 = tomo_mm(dRz, Rx), A4 = tomo_mm(Rz, dRx), A5 = tomo_mm(Rzx, dRy);
        TomoM3 Rab = tomo_mm(Rx, Ry);
        const double rv[3] = {0.0, g.det_y - g.src_y, 0.0};         // untransformed ray  :158
        double tmp[3], tmp2[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                gc->rzx[i][j] = Rzx.m[i][j];
                gc->a3[i][j] = A3.m[i][j];
                gc->a4[i][j] = A4.m[i][j];
                gc->a5[i][j] = A5.m[i][j];
                gc->ry[i][j] = Ry.m[i][j];
            }
        for (int a = 0; a < 3; ++a) { gc->t[a] = t[a]; gc->s00[a] = s00[a]; }
        gc->sdx = g.det_dx;
        gc->sdz = g.det_dz;
        tomo_mv(Rab, rv, tmp);  tomo_mv(dRz, tmp, gc->app[0]);      // :47
        tomo_mv(Ry, rv, tmp);   tomo_mv(dRx, tmp, tmp2); tomo_mv(Rz, tmp2, gc->app[1]);  // :48
        tomo_mv(dRy, rv, tmp);  tomo_mv(Rzx, tmp, gc->app[2]);      // :49
    }
}

// {t : lo <= b + t*d < hi} intersected into [t0, t1]
TOMO_HD void tomo_clip_axis(double b, double d, double lo, double hi, double &t0, double &t1)
{
    if (d != 0.0) {
        double ta = (lo - b) / d, tb = (hi - b) / d;
        if (ta > tb) { double s = ta; ta = tb; tb = s; }
        t0 = t0 > ta ? t0 : ta;
        t1 = t1 < tb ? t1 : tb;
    } else if (b < lo || b >= hi) {
        t0 = 1.0;
        t1 = 0.0;
    }
}

// Sample range [j0, j1) of a ray outside which every corner of every sample is out of bounds
// (floor(p_a) in [-1, n_a-1] <=> p_a in [-1, n_a)).  Rounded outward by 1e-6 samples only, so all
// computed floors stay inside the TOMO_HALO=2 padding (float32 in-block error ~1e-6 voxel).
// With a box [lo_a, hi_a) of the NON-ZERO voxels (found while the volume is staged, k_pad) the range shrinks to the samples
// that can see one of them: floor(p_a) in [lo_a - 1, hi_a - 1] <=> p_a in [lo_a - 1, hi_a) -- the others add exactly 0 to a
// projection and to its gradient (all 8 corners are 0), the ray-driven analogue of the tile kernels' all-zero-tile exit.
TOMO_HD void tomo_ray_range_box(const double b[3], const double d[3], int n, const int lo[3], const int hi[3], int &j0, int &j1)
{
    j0 = j1 = 0;
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) return;      // nothing non-zero
    double t0 = 0.0, t1 = (double)(n - 1);
    tomo_clip_axis(b[0], d[0], (double)lo[0] - 1.0, (double)hi[0], t0, t1);
    tomo_clip_axis(b[1], d[1], (double)lo[1] - 1.0, (double)hi[1], t0, t1);
    tomo_clip_axis(b[2], d[2], (double)lo[2] - 1.0, (double)hi[2], t0, t1);
    if (!(t0 <= t1)) return;
    j0 = (int)ceil(t0 - 1e-6);
    j1 = (int)floor(t1 + 1e-6) + 1;
    if (j0 < 0) j0 = 0;
    if (j1 > n) j1 = n;
    if (j1 < j0) j1 = j0;
}
TOMO_HD void tomo_ray_range(const double b[3], const double d[3], int n, int nx, int ny, int nz, int &j0, int &j1)
{
    const int lo[3] = {0, 0, 0}, hi[3] = {nx, ny, nz};
    tomo_ray_range_box(b, d, n, lo, hi, j0, j1);
}

// Block anchor: the integer floor of the block's MIDDLE position, so the float32 in-block coordinates x = f0 + jj*d stay within
// +-(TOMO_JB/2 * |d| + 1) <= +-17 FOR A SAMPLE STEP <= 1 VOXEL (tomo_check_geometry routes step > 1 to the plain 64-bit kernels,
// which call this with the same arithmetic but do not rely on the range; their accuracy degrades as step * 1e-6): float32 rounding of a sample position <= ~1e-6 voxel (f0 and the fma each round at
// ulp(16) = 1.9e-6 at the block ends, 2.4e-7 in the middle) -- half of what an anchor below the block's lowest corner gave
// (coordinates up to 35), which at 512^3 left the pose gradient only just inside 1e-5 of the float64 reference on a
// piecewise-constant phantom (tests/test_gpu_configs.py).  Cells relative to the anchor lie in [-TOMO_ABIAS + 1, TOMO_ABIAS - 1]:
// the kernels that form UNSIGNED 32-bit byte offsets lower their wave-uniform base pointer by TOMO_ABIAS cells per axis and
// add the same to the lane offset (tomo_abias_bytes).
#define TOMO_ABIAS 18
TOMO_HD void tomo_block_anchor(const double b[3], const double d[3], int jb, int ia[3], float f0[3], int span = TOMO_JB)
{
    for (int a = 0; a < 3; ++a) {
        double s = b[a] + (double)jb * d[a];
        double f = floor(s + 0.5 * (double)(span - 1) * d[a]);
        ia[a] = (int)f;
        f0[a] = (float)(s - f);
    }
}
TOMO_HD uint32_t tomo_abias_bytes(uint32_t sx4, uint32_t sy4) { return (uint32_t)TOMO_ABIAS * (sx4 + sy4 + 4u); }
// The lanes of a wave are 64 consecutive detector-z rays: their anchors differ from lane 0's by at most 64 |w_a| + 1 <= 65 cells
// per axis (|w| = detector-z pitch in voxels <= 1: checked when the poses are staged), so a fixed bias of 66 cells per axis makes
// every lane's offset relative to lane 0's anchor non-negative -- no per-block wave-wide minimum (six cross-lane round trips).
#define TOMO_LBIAS 66
TOMO_HD uint32_t tomo_lbias_bytes(uint32_t sx4, uint32_t sy4) { return (uint32_t)TOMO_LBIAS * (sx4 + sy4 + 4u); }

#endif  // TOMO_RAYCORE_H_
