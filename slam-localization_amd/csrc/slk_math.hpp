// slk_math.hpp -- device-side manifold arithmetic and the registered models.
//
// SO(3) follows MTK::SO3<double> (third-party of the reference, restated from its published
// algorithm): exp(v) = (sinc(|v|/2)/2 * v, cos(|v|/2)), log(q) = 2 atan(|vec|/w)/|vec| * vec,
// boxplus q <- q * exp(v), boxminus log(other^-1 * q).  Used by reference
// src/filters/State.hpp:166-200 (State::set / boxplus / boxminus) and :215-239.
#pragma once
#include <hip/hip_runtime.h>

namespace slk {

struct Quat { double x, y, z, w; };

__device__ __forceinline__ Quat qmul(const Quat &a, const Quat &b)
{
    Quat o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    o.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return o;
}

__device__ __forceinline__ Quat qconj(const Quat &q) { return Quat{-q.x, -q.y, -q.z, q.w}; }

// Eigen QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v)
__device__ __forceinline__ void qrot(const Quat &q, double vx, double vy, double vz,
                                     double &ox, double &oy, double &oz)
{
    double ux = q.y * vz - q.z * vy, uy = q.z * vx - q.x * vz, uz = q.x * vy - q.y * vx;
    ux += ux; uy += uy; uz += uz;
    ox = vx + q.w * ux + (q.y * uz - q.z * uy);
    oy = vy + q.w * uy + (q.z * ux - q.x * uz);
    oz = vz + q.w * uz + (q.x * uy - q.y * ux);
}

// MTK cos_sinc_sqrt(x) = (cos(sqrt x), sin(sqrt x)/sqrt x).  Both are entire functions of x:
//   cos(sqrt x) = sum (-x)^k/(2k)!,   sin(sqrt x)/sqrt x = sum (-x)^k/(2k+1)!
// For x < 1/4 (rotation below 1 rad, the usual sigma-point spread) polynomials of degree 6 / 5 in x stand for the
// series (near-minimax, below 1 ulp) -- no sqrt, no division, no range reduction; MTK itself switches to
// this series for tiny x (3 terms below eps^(1/4)).  Larger arguments take the libm route.
// The libm routes are rare (rotations beyond the series' domain) and register-hungry: kept out of line so that
// their temporaries and polynomial constants do not count against the callers' register budget.
// (results by value: a local whose address goes to an out-of-line function would live in scratch memory)
struct CosSinc { double c, s; };
__device__ __forceinline__ CosSinc cos_sinc_sqrt_libm_body(double x)
{
    const double sx = sqrt(x);
    return CosSinc{cos(sx), sin(sx) / sx};
}
__device__ __forceinline__ double so3_log_scale_libm_body(double n2, double w)
{
    double nv = sqrt(n2);
    if (nv < 1e-11) nv = 1e-11;
    return 2.0 / nv * atan(nv / w);
}
__device__ __attribute__((noinline)) CosSinc cos_sinc_sqrt_libm(double x) { return cos_sinc_sqrt_libm_body(x); }
__device__ __attribute__((noinline)) double so3_log_scale_libm(double n2, double w) { return so3_log_scale_libm_body(n2, w); }

// LEAF = the libm route inlined (for callers that are out-of-line functions themselves and must stay leaves)
template <bool LEAF = false>
__device__ __forceinline__ void cos_sinc_sqrt(double x, double &c, double &s)
{
    if (x < 0.25) {
        const double y = -x;                      // Horner in y = -x; near-minimax coefficients (tools/series_coefficients.py):
        double cc = 0x1.1d8d32755f8fbp-29;     // cos sqrt x, degree 6: relative error 6e-18 on x <= 1/4
        cc = fma(cc, y, 0x1.27e40964b47d4p-22);
        cc = fma(cc, y, 0x1.a01a00fb1bc6fp-16);
        cc = fma(cc, y, 0x1.6c16c16bdd04ep-10);
        cc = fma(cc, y, 0x1.5555555555421p-5);
        cc = fma(cc, y, 0x1.fffffffffffffp-2);
        cc = fma(cc, y, 1.0);
        double ss = 0x1.ac53ce336f805p-26;     // sin sqrt x / sqrt x, degree 5: 4e-17
        ss = fma(ss, y, 0x1.71dd113fb7905p-19);
        ss = fma(ss, y, 0x1.a01a01061c190p-13);
        ss = fma(ss, y, 0x1.11111110ecfb3p-7);
        ss = fma(ss, y, 0x1.555555555548fp-3);
        ss = fma(ss, y, 1.0);
        c = cc;
        s = ss;
    } else {
        const CosSinc r = LEAF ? cos_sinc_sqrt_libm_body(x) : cos_sinc_sqrt_libm(x);
        c = r.c;
        s = r.s;
    }
}

template <bool LEAF = false>
__device__ __forceinline__ Quat so3_exp(double vx, double vy, double vz)
{
    double c, s;
    cos_sinc_sqrt<LEAF>(0.25 * (vx * vx + vy * vy + vz * vz), c, s);
    double m = s * 0.5;
    return Quat{m * vx, m * vy, m * vz, c};
}

// MTK::SO3::log: 2 atan(|vec|/w)/|vec| * vec (|vec| clamped to 1e-11).  With u = |vec|/w the factor is
// 2/w * atan(u)/u and atan(u)/u = sum (-u^2)^k/(2k+1): for u^2 < 1/16 (rotation below ~28 deg) the
// series is replaced by a polynomial of degree 8 in u^2 (near-minimax, < 1 ulp) -- one reciprocal instead of sqrt + 2 divisions + atan.
template <bool LEAF = false>
__device__ __forceinline__ void so3_log(const Quat &q, double &vx, double &vy, double &vz)
{
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
    const double w2 = q.w * q.w;
    double s;
    if (q.w > 0.0 && n2 * 16.0 < w2) {
        // 1 / w by v_rcp_f64 and two Newton steps (to about an ulp) instead of the IEEE division sequence
        double rw = __builtin_amdgcn_rcp(q.w);
        rw = fma(fma(-q.w, rw, 1.0), rw, rw);
        rw = fma(fma(-q.w, rw, 1.0), rw, rw);
        const double y = -(n2 * rw * rw);
        double f = 0x1.78be0a9b1dd1fp-5;          // atan(u) / u, degree 8 in y: near-minimax (tools/series_coefficients.py), relative error 9e-18
        f = fma(f, y, 0x1.0b3340fe2ed9ap-4);
        f = fma(f, y, 0x1.3ab708d770276p-4);
        f = fma(f, y, 0x1.7459b99bfc19bp-4);
        f = fma(f, y, 0x1.c71c5f4b9c2adp-4);
        f = fma(f, y, 0x1.24924907fa636p-3);
        f = fma(f, y, 0x1.999999996d307p-3);
        f = fma(f, y, 0x1.5555555555481p-2);
        f = fma(f, y, 1.0);
        s = 2.0 * f * rw;
    } else {
        s = LEAF ? so3_log_scale_libm_body(n2, q.w) : so3_log_scale_libm(n2, q.w);
    }
    vx = s * q.x; vy = s * q.y; vz = s * q.z;
}

// other [-] ... : log(b^-1 * a)
__device__ __forceinline__ void so3_boxminus(const Quat &a, const Quat &b, double &vx, double &vy, double &vz)
{
    so3_log(qmul(qconj(b), a), vx, vy, vz);
}

__device__ __forceinline__ Quat ldq(const double *p) { return Quat{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void stq(double *p, const Quat &q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }

// ---------------------------------------------------------------- single State (13) helpers
// x [+] v for one State (State.hpp:186-192); v has 12 entries
__device__ __forceinline__ void state_boxplus(const double *x, const double *v, double *o)
{
    o[0] = x[0] + v[0]; o[1] = x[1] + v[1]; o[2] = x[2] + v[2];
    stq(o + 3, qmul(ldq(x + 3), so3_exp(v[3], v[4], v[5])));
    o[7] = x[7] + v[6]; o[8] = x[8] + v[7]; o[9] = x[9] + v[8];
    o[10] = x[10] + v[9]; o[11] = x[11] + v[10]; o[12] = x[12] + v[11];
}

// a [-] b for one State (State.hpp:194-200)
__device__ __forceinline__ void state_boxminus(const double *a, const double *b, double *d)
{
    d[0] = a[0] - b[0]; d[1] = a[1] - b[1]; d[2] = a[2] - b[2];
    so3_boxminus(ldq(a + 3), ldq(b + 3), d[3], d[4], d[5]);
    d[6] = a[7] - b[7]; d[7] = a[8] - b[8]; d[8] = a[9] - b[9];
    d[9] = a[10] - b[10]; d[10] = a[11] - b[11]; d[11] = a[12] - b[12];
}

// ---------------------------------------------------------------- dead reckoning
// DeadReckon::updateAttitude (src/core/DeadReckon.hpp:246-286): third-order quaternion integration, constant angular
// acceleration between the previous (w1) and the current (w0) sample.  The reference's 4x4 expression acts on the
// identity quaternion, i.e. only its first column is used: omega4 e0 = (0, w), (omega4 oldomega4) e0 = (-w0.w1, -(w0 x w1)).
__device__ __forceinline__ Quat update_attitude(double dt, const double *w0, const double *w1)
{
    const double n2 = w0[0] * w0[0] + w0[1] * w0[1] + w0[2] * w0[2];
    const double dot = w0[0] * w1[0] + w0[1] * w1[1] + w0[2] * w1[2];
    const double cx = w0[1] * w1[2] - w0[2] * w1[1], cy = w0[2] * w1[0] - w0[0] * w1[2], cz = w0[0] * w1[1] - w0[1] * w1[0];
    const double dt2 = dt * dt, dt3 = dt2 * dt;
    const double qw = 1.0 - (1.0 / 6.0) * n2 * dt2 + (1.0 / 24.0) * dot * dt2;
    const double k = (1.0 / 48.0) * n2 * dt3;
    const double qx = 0.75 * w0[0] * dt - 0.25 * w1[0] * dt + (1.0 / 24.0) * cx * dt2 - k * w0[0];
    const double qy = 0.75 * w0[1] * dt - 0.25 * w1[1] * dt + (1.0 / 24.0) * cy * dt2 - k * w0[1];
    const double qz = 0.75 * w0[2] * dt - 0.25 * w1[2] * dt + (1.0 / 24.0) * cz * dt2 - k * w0[2];
    const double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    return Quat{qx / nrm, qy / nrm, qz / nrm, qw / nrm};
}
// DeadReckon::updatePose delta pose (src/core/DeadReckon.hpp:129-239): u = dt v0[3] w0[3] v1[3] w1[3] ->
// d = dpos[3] dquat[4] velocity[3] angular_velocity[3] (the input of the delta-pose process model)
__device__ __forceinline__ void dead_reckon_delta(const double *u, double *d)
{
    const double dt = u[0];
    for (int i = 0; i < 3; ++i) d[i] = (dt / 2.0) * (u[1 + i] + u[7 + i]);
    stq(d + 3, update_attitude(dt, u + 4, u + 10));
    for (int i = 0; i < 3; ++i) { d[7 + i] = u[1 + i]; d[10 + i] = u[4 + i]; }
}

// ---------------------------------------------------------------- registered process models
// SLK_PM_CONST_VELOCITY: test/UsckfUnitTest.cpp:34-49;  u = v[3] w[3] dt
// SLK_PM_DELTA_POSE:     test/MsckfUnitTest.cpp:33-47;  u = dpos[3] dquat[4] v[3] w[3]
// SLK_PM_DEAD_RECKON:    src/core/DeadReckon.hpp:129-239 feeding the delta-pose model; u = dt v0[3] w0[3] v1[3] w1[3]
__device__ __forceinline__ void process_model(int model, const double *u, const double *x, double *y)
{
    double uu[13];              // the inputs in registers (a pointer that may address either global memory or a local
                                // array would force the array into scratch memory)
    if (model == 3) {           // SLK_PM_DEAD_RECKON: the delta pose comes from the velocity samples
        dead_reckon_delta(u, uu);
    } else {
#pragma unroll
        for (int i = 0; i < 13; ++i) uu[i] = (model == 1 && i > 6) ? 0.0 : u[i];
    }
    u = nullptr;
    if (model == 1) {
        double dt = uu[6];
        Quat rot = so3_exp(uu[3] * dt, uu[4] * dt, uu[5] * dt);
        stq(y + 3, qmul(ldq(x + 3), rot));
        y[10] = uu[3]; y[11] = uu[4]; y[12] = uu[5];
        y[7] = uu[0]; y[8] = uu[1]; y[9] = uu[2];
        y[0] = x[0] + x[7] * dt; y[1] = x[1] + x[8] * dt; y[2] = x[2] + x[9] * dt;
    } else {
        Quat q = qmul(ldq(x + 3), ldq(uu + 3));
        stq(y + 3, q);
        y[10] = uu[10]; y[11] = uu[11]; y[12] = uu[12];
        double rx, ry, rz;
        qrot(q, uu[0], uu[1], uu[2], rx, ry, rz);
        y[0] = x[0] + rx; y[1] = x[1] + ry; y[2] = x[2] + rz;
        y[7] = uu[7]; y[8] = uu[8]; y[9] = uu[9];
    }
}

// Eigen toRotationMatrix (through Eigen::Affine3d(orient), UsckfUnitTest.cpp:71), row-wise apply
__device__ __forceinline__ void qmat_apply(const Quat &q, double cx, double cy, double cz,
                                           double &ox, double &oy, double &oz)
{
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    ox = (1 - (tyy + tzz)) * cx + (txy - twz) * cy + (txz + twy) * cz;
    oy = (txy + twz) * cx + (1 - (txx + tzz)) * cy + (tyz - twx) * cz;
    oz = (txz - twy) * cx + (tyz + twx) * cy + (1 - (txx + tyy)) * cz;
}

} // namespace slk
