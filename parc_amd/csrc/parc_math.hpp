// parc_math.hpp — device-side fp32 quaternion / exp-map algebra for the gfx950 kernels.
//
// Each function follows the operation order of the reference's TorchScript op of the same name in
// PARC/util/torch_util.py (file:line in the comments, relative to /root/reference; semantics table in
// SURVEY.md Appendix A) so that results stay within 1e-5 of the PyTorch path.  The translation unit is
// compiled with -ffp-contract=off: a fused multiply-add would round differently from ATen's separate
// multiply and add, which matters for the truncations (frame index, grid cell) further down the pipe.
#pragma once
#include <hip/hip_runtime.h>

namespace parc {

struct V3 { float x, y, z; };
typedef float4 Q4; // (x, y, z, w)

__device__ __forceinline__ V3 mk3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ Q4 mk4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }

__device__ __forceinline__ float norm3(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

// torch_util.py:11-13  x / clamp(||x||, min=1e-9)
__device__ __forceinline__ V3 normalize3(V3 v) {
    float n = fmaxf(norm3(v), 1e-9f);
    return mk3(v.x / n, v.y / n, v.z / n);
}
__device__ __forceinline__ Q4 normalize4(Q4 q) {
    float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return mk4(q.x / n, q.y / n, q.z / n, q.w / n);
}

// torch_util.py:42-59 — the 9-multiply factored product (NOT the textbook 16-multiply form)
__device__ __forceinline__ Q4 quat_mul(Q4 a, Q4 b) {
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    float w = qq - ww + (a.z - a.y) * (b.y - b.z);
    float x = qq - xx + (a.x + a.w) * (b.x + b.w);
    float y = qq - yy + (a.w - a.x) * (b.y + b.z);
    float z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return mk4(x, y, z, w);
}

__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// torch_util.py:62-67  t = 2 (q_v x v);  v + q_w t + q_v x t
__device__ __forceinline__ V3 quat_rotate(Q4 q, V3 v) {
    V3 qv = mk3(q.x, q.y, q.z);
    V3 t = cross3(qv, v);
    t.x = 2.f * t.x; t.y = 2.f * t.y; t.z = 2.f * t.z;
    V3 c = cross3(qv, t);
    return mk3(v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z);
}

__device__ __forceinline__ Q4 quat_conj(Q4 q) { return mk4(-q.x, -q.y, -q.z, q.w); }

// torch_util.py:35-39
__device__ __forceinline__ Q4 quat_pos(Q4 q) {
    float s = 1.f - 2.f * ((q.w < 0.f) ? 1.f : 0.f);
    return mk4(s * q.x, s * q.y, s * q.z, s * q.w);
}

// torch_util.py:70-91
__device__ __forceinline__ void quat_to_axis_angle(Q4 qin, V3 &axis, float &angle) {
    Q4 q = quat_pos(qin);
    float len = norm3(mk3(q.x, q.y, q.z));
    float ang = 2.0f * atan2f(len, q.w);
    float safe = fmaxf(len, 1e-6f);
    bool ok = len > 1e-5f;
    axis = ok ? mk3(q.x / safe, q.y / safe, q.z / safe) : mk3(0.f, 0.f, 1.f);
    angle = ok ? ang : 0.f;
}

// torch_util.py:337-342
__device__ __forceinline__ Q4 axis_angle_to_quat(V3 axis, float angle) {
    float theta = angle / 2.f;
    V3 na = normalize3(axis);
    float s = sinf(theta), c = cosf(theta);
    return normalize4(mk4(na.x * s, na.y * s, na.z * s, c));
}

// torch_util.py:426-450 — divides by a possibly-zero angle before masking (inf/NaN masked after)
__device__ __forceinline__ Q4 exp_map_to_quat(V3 e) {
    float ang = norm3(e);
    V3 ax = mk3(e.x / ang, e.y / ang, e.z / ang);
    ang = atan2f(sinf(ang), cosf(ang)); // normalize_angle :6-8
    bool ok = fabsf(ang) > 1e-5f;
    V3 axis = ok ? ax : mk3(0.f, 0.f, 1.f);
    return axis_angle_to_quat(axis, ok ? ang : 0.f);
}

// torch_util.py:372-376
__device__ __forceinline__ V3 quat_to_exp_map(Q4 q) {
    V3 axis; float angle;
    quat_to_axis_angle(q, axis, angle);
    return mk3(angle * axis.x, angle * axis.y, angle * axis.z);
}

// torch_util.py:454-462  angle of q1 (x) conj(q0)
__device__ __forceinline__ float quat_diff_angle(Q4 q0, Q4 q1) {
    V3 axis; float angle;
    quat_to_axis_angle(quat_mul(q1, quat_conj(q0)), axis, angle);
    return angle;
}

// torch_util.py:470-472
__device__ __forceinline__ Q4 quat_normalize(Q4 q) { return normalize4(quat_pos(q)); }

// quat_rotate(q, (1,0,0)) and quat_rotate(q, (0,0,1)) with the multiplications by the 0/1 components of v
// carried out symbolically: x*0 = +-0 and a + (+-0) = a, so every surviving operation is the one the generic
// formula performs, in the same order — results are identical up to the sign of an exact zero.
__device__ __forceinline__ V3 quat_rotate_x(Q4 q) { // v = (1,0,0): t = 2(0, qz, -qy)
    const float ty = 2.f * q.z, tz = 2.f * (-q.y);
    const float cx = q.y * tz - q.z * ty, cy = -(q.x * tz), cz = q.x * ty;
    return mk3(1.f + cx, q.w * ty + cy, q.w * tz + cz);
}
__device__ __forceinline__ V3 quat_rotate_z(Q4 q) { // v = (0,0,1): t = 2(qy, -qx, 0)
    const float tx = 2.f * q.y, ty = 2.f * (-q.x);
    const float cx = -(q.z * ty), cy = q.z * tx, cz = q.x * ty - q.y * tx;
    return mk3(q.w * tx + cx, q.w * ty + cy, 1.f + cz);
}

// torch_util.py:393-404: [rot(q,(1,0,0)), rot(q,(0,0,1))]
__device__ __forceinline__ void quat_to_tan_norm(Q4 q, float *o) {
    V3 t = quat_rotate_x(q);
    V3 n = quat_rotate_z(q);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
}

// torch_util.py:475-499 — no renormalisation; fall-backs |sin|<1e-3 -> midpoint, |cos|>=1 -> q0
__device__ __forceinline__ Q4 slerp(Q4 q0, Q4 q1, float t) {
    float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (c < 0.f) q1 = mk4(-q1.x, -q1.y, -q1.z, -q1.w);
    c = fabsf(c);
    float h = acosf(c);
    float s = sqrtf(1.0f - c * c);
    float ra = sinf((1.f - t) * h) / s;
    float rb = sinf(t * h) / s;
    Q4 r = mk4(ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w);
    if (fabsf(s) < 0.001f)
        r = mk4(0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w);
    if (fabsf(c) >= 1.f) r = q0;
    return r;
}

// ---- slerp with reduced-range transcendentals (the observation kernel's 105 slerps per env-step) -------------------------
// The arguments of the reference's formula never leave [0, 1] for acos (|cos| of the half angle) and [0, pi/2] for sin
// ((1-t) h and t h with t in [0, 1]), so the general-purpose routines' range reduction and special cases are dead weight:
// two short polynomials do (each within 1.4e-7 relative of the exact function over the whole range: fitted and checked in
// fp32 by tests/test_host_cpu.py; the device result is compared with the reference's slerp rows and with slerp() above by the
// GPU tests).  Same formula, same fall-backs; ~40 instructions instead of ~170.  NOT bit-identical to slerp(): the reset path
// and the stand-alone ops keep the accurate version.
__device__ __forceinline__ float sin_q1(float x) { // x in [0, 1.62]
    const float z = x * x;
    float p = -2.4019863431590238e-08f;
    p = fmaf(p, z, 2.753377657427336e-06f);
    p = fmaf(p, z, -0.00019841050379909575f);
    p = fmaf(p, z, 0.00833333283662796f);
    p = fmaf(p, z, -0.1666666716337204f);
    return fmaf(x * z, p, x);
}
__device__ __forceinline__ float acos_01(float c) { // c in [0, 1]
    const bool big = c > 0.5f;                      // acos c = 2 asin sqrt((1-c)/2) there, pi/2 - asin c below
    const float zb = (1.0f - c) * 0.5f;
    const float z = big ? zb : c * c;
    const float u = big ? __builtin_amdgcn_sqrtf(zb) : c;
    float p = 0.03385632857680321f;                 // asin u = u + u z R(z), z = u^2 <= 0.25
    p = fmaf(p, z, 0.01704932190477848f);
    p = fmaf(p, z, 0.03112172894179821f);
    p = fmaf(p, z, 0.04459759593009949f);
    p = fmaf(p, z, 0.07500100135803223f);
    p = fmaf(p, z, 0.1666666567325592f);
    const float a = fmaf(u * z, p, u);
    return big ? 2.0f * a : 1.5707963267948966f - a;
}
__device__ __forceinline__ Q4 slerp_rr(Q4 q0, Q4 q1, float t) {
    float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (c < 0.f) q1 = mk4(-q1.x, -q1.y, -q1.z, -q1.w);
    c = fabsf(c);
    const float cc = fminf(c, 1.0f);                // rounding can leave |c| a few ulp above 1: the last fall-back takes those
    const float h = acos_01(cc);
    const float s = __builtin_amdgcn_sqrtf(fmaxf(1.0f - cc * cc, 0.f));
    const float inv = __builtin_amdgcn_rcpf(s);
    const float ra = sin_q1((1.f - t) * h) * inv;
    const float rb = sin_q1(t * h) * inv;
    Q4 r = mk4(ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w);
    if (fabsf(s) < 0.001f)
        r = mk4(0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w);
    if (fabsf(c) >= 1.f) r = q0;
    return r;
}

// torch_util.py:502-511
__device__ __forceinline__ float calc_heading(Q4 q) {
    V3 d = quat_rotate_x(q);
    return atan2f(d.y, d.x);
}

// torch_util.py:523-530
__device__ __forceinline__ Q4 heading_quat_inv(float heading) { return axis_angle_to_quat(mk3(0.f, 0.f, 1.f), -heading); }

} // namespace parc
