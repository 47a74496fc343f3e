/*
 * parc_oracle.c — see parc_oracle.h.  TEST INFRASTRUCTURE: never part of the product path.
 *
 * Scalar fp32 restatement of the reference's PyTorch ops, one element / one env at a time, in the
 * reference's operation order.  Build with -ffp-contract=off so that every
 * multiply and add rounds separately, like the ATen element-wise kernels it restates.
 * file:line citations are relative to /root/reference.
 */
#include "parc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * a1 — PARC/util/torch_util.py
 * ---------------------------------------------------------------------------------------- */
static inline float norm3(const float *v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static inline float norm4(const float *v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]); }

/* torch_util.py:11-13  x / clamp(||x||, min=1e-9) */
static inline void normalize_n(const float *x, float *o, int dim) {
    float s = 0.f;
    for (int i = 0; i < dim; ++i) s = s + x[i] * x[i];
    float n = sqrtf(s);
    if (n < 1e-9f) n = 1e-9f;
    for (int i = 0; i < dim; ++i) o[i] = x[i] / n;
}

/* torch_util.py:42-59 — 9-multiply factored form */
static inline void quat_mul(const float *a, const float *b, float *o) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float w = qq - ww + (z1 - y1) * (y2 - z2);
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* torch_util.py:608-626 — textbook form */
static inline void quat_multiply(const float *q1, const float *q2, float *o) {
    float x1 = q1[0], y1 = q1[1], z1 = q1[2], w1 = q1[3];
    float x2 = q2[0], y2 = q2[1], z2 = q2[2], w2 = q2[3];
    o[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    o[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    o[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    o[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
}

static inline void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* torch_util.py:62-67 */
static inline void quat_rotate(const float *q, const float *v, float *o) {
    float t[3], c[3];
    cross3(q, v, t);
    t[0] = 2.f * t[0]; t[1] = 2.f * t[1]; t[2] = 2.f * t[2];
    cross3(q, t, c);
    float r0 = v[0] + q[3] * t[0] + c[0];
    float r1 = v[1] + q[3] * t[1] + c[1];
    float r2 = v[2] + q[3] * t[2] + c[2];
    o[0] = r0; o[1] = r1; o[2] = r2;
}

static inline void quat_conjugate(const float *q, float *o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }

/* torch_util.py:35-39 */
static inline void quat_pos(const float *q, float *o) {
    float z = (q[3] < 0.f) ? 1.f : 0.f;
    float s = 1.f - 2.f * z;
    o[0] = s * q[0]; o[1] = s * q[1]; o[2] = s * q[2]; o[3] = s * q[3];
}

/* torch_util.py:70-91 */
static inline void quat_to_axis_angle(const float *q_in, float *axis, float *angle) {
    float q[4];
    quat_pos(q_in, q);
    float length = norm3(q);
    float ang = 2.0f * atan2f(length, q[3]);
    float safe = length < 1e-6f ? 1e-6f : length;
    float ax0 = q[0] / safe, ax1 = q[1] / safe, ax2 = q[2] / safe;
    if (length > 1e-5f) { axis[0] = ax0; axis[1] = ax1; axis[2] = ax2; *angle = ang; }
    else { axis[0] = 0.f; axis[1] = 0.f; axis[2] = 1.f; *angle = 0.f; }
}

/* torch_util.py:337-342 */
static inline void axis_angle_to_quat(const float *axis, float angle, float *o) {
    float theta = angle / 2.f;
    float na[3], q[4];
    normalize_n(axis, na, 3);
    float s = sinf(theta);
    q[0] = na[0] * s; q[1] = na[1] * s; q[2] = na[2] * s; q[3] = cosf(theta);
    normalize_n(q, o, 4);
}

/* torch_util.py:6-8 */
static inline float normalize_angle(float x) { return atan2f(sinf(x), cosf(x)); }

/* torch_util.py:426-443 — divides by a possibly-zero angle BEFORE masking */
static inline void exp_map_to_axis_angle(const float *e, float *axis, float *angle) {
    float ang = norm3(e);
    float ax0 = e[0] / ang, ax1 = e[1] / ang, ax2 = e[2] / ang;
    ang = normalize_angle(ang);
    if (fabsf(ang) > 1e-5f) { axis[0] = ax0; axis[1] = ax1; axis[2] = ax2; *angle = ang; }
    else { axis[0] = 0.f; axis[1] = 0.f; axis[2] = 1.f; *angle = 0.f; }
}

/* torch_util.py:446-450 */
static inline void exp_map_to_quat(const float *e, float *o) {
    float axis[3], angle;
    exp_map_to_axis_angle(e, axis, &angle);
    axis_angle_to_quat(axis, angle, o);
}

/* torch_util.py:372-376 (+ axis_angle_to_exp_map:356) */
static inline void quat_to_exp_map(const float *q, float *o) {
    float axis[3], angle;
    quat_to_axis_angle(q, axis, &angle);
    o[0] = angle * axis[0]; o[1] = angle * axis[1]; o[2] = angle * axis[2];
}

/* torch_util.py:454-456  q1 (x) conj(q0) */
static inline void quat_diff(const float *q0, const float *q1, float *o) {
    float c[4];
    quat_conjugate(q0, c);
    quat_mul(q1, c, o);
}

/* torch_util.py:459-462 */
static inline float quat_diff_angle(const float *q0, const float *q1) {
    float d[4], axis[3], angle;
    quat_diff(q0, q1, d);
    quat_to_axis_angle(d, axis, &angle);
    return angle;
}

/* torch_util.py:470-472 */
static inline void quat_normalize(const float *q, float *o) {
    float p[4];
    quat_pos(q, p);
    normalize_n(p, o, 4);
}

/* torch_util.py:393-404 */
static inline void quat_to_tan_norm(const float *q, float *o) {
    const float tx[3] = {1.f, 0.f, 0.f}, nz[3] = {0.f, 0.f, 1.f};
    quat_rotate(q, tx, o);
    quat_rotate(q, nz, o + 3);
}

/* torch_util.py:475-499 — no renormalisation; two fall-backs applied in this order */
static inline void slerp(const float *q0, const float *q1_in, float t, float *o) {
    float c = q0[0] * q1_in[0] + q0[1] * q1_in[1] + q0[2] * q1_in[2] + q0[3] * q1_in[3];
    float q1[4];
    if (c < 0.f) { q1[0] = -q1_in[0]; q1[1] = -q1_in[1]; q1[2] = -q1_in[2]; q1[3] = -q1_in[3]; }
    else { q1[0] = q1_in[0]; q1[1] = q1_in[1]; q1[2] = q1_in[2]; q1[3] = q1_in[3]; }
    c = fabsf(c);
    float half_theta = acosf(c);
    float s = sqrtf(1.0f - c * c);
    float ra = sinf((1.f - t) * half_theta) / s;
    float rb = sinf(t * half_theta) / s;
    for (int i = 0; i < 4; ++i) {
        float v = ra * q0[i] + rb * q1[i];
        if (fabsf(s) < 0.001f) v = 0.5f * q0[i] + 0.5f * q1[i];
        if (fabsf(c) >= 1.f) v = q0[i];
        o[i] = v;
    }
}

/* torch_util.py:502-511 */
static inline float calc_heading(const float *q) {
    const float ref[3] = {1.f, 0.f, 0.f};
    float d[3];
    quat_rotate(q, ref, d);
    return atan2f(d[1], d[0]);
}

/* torch_util.py:523-530 */
static inline void calc_heading_quat_inv(const float *q, float *o) {
    const float z[3] = {0.f, 0.f, 1.f};
    axis_angle_to_quat(z, -calc_heading(q), o);
}

/* torch_util.py:651-663 */
static inline void rotate_2d_vec(const float *v, float angle, float *o) {
    float c = cosf(angle), s = sinf(angle);
    float rx = v[0] * c - v[1] * s;
    float ry = v[0] * s + v[1] * c;
    o[0] = rx; o[1] = ry;
}

#define BATCH1(name, fn, in_dim, out_dim) \
    void name(const float *a, float *out, int n) { for (int i = 0; i < n; ++i) fn(a + (size_t)i * in_dim, out + (size_t)i * out_dim); }
#define BATCH2(name, fn, d1, d2, out_dim) \
    void name(const float *a, const float *b, float *out, int n) { for (int i = 0; i < n; ++i) fn(a + (size_t)i * d1, b + (size_t)i * d2, out + (size_t)i * out_dim); }

BATCH2(orc_quat_mul, quat_mul, 4, 4, 4)
BATCH2(orc_quat_multiply, quat_multiply, 4, 4, 4)
BATCH2(orc_quat_rotate, quat_rotate, 4, 3, 3)
BATCH1(orc_quat_conjugate, quat_conjugate, 4, 4)
BATCH1(orc_quat_pos, quat_pos, 4, 4)
BATCH1(orc_exp_map_to_quat, exp_map_to_quat, 3, 4)
BATCH1(orc_quat_to_exp_map, quat_to_exp_map, 4, 3)
BATCH2(orc_quat_diff, quat_diff, 4, 4, 4)
BATCH1(orc_quat_normalize, quat_normalize, 4, 4)
BATCH1(orc_quat_to_tan_norm, quat_to_tan_norm, 4, 6)
BATCH1(orc_calc_heading_quat_inv, calc_heading_quat_inv, 4, 4)

void orc_normalize(const float *x, float *out, int n, int dim) { for (int i = 0; i < n; ++i) normalize_n(x + (size_t)i * dim, out + (size_t)i * dim, dim); }
void orc_quat_to_axis_angle(const float *q, float *axis, float *angle, int n) { for (int i = 0; i < n; ++i) quat_to_axis_angle(q + 4 * i, axis + 3 * i, angle + i); }
void orc_axis_angle_to_quat(const float *axis, const float *angle, float *out, int n) { for (int i = 0; i < n; ++i) axis_angle_to_quat(axis + 3 * i, angle[i], out + 4 * i); }
void orc_exp_map_to_axis_angle(const float *e, float *axis, float *angle, int n) { for (int i = 0; i < n; ++i) exp_map_to_axis_angle(e + 3 * i, axis + 3 * i, angle + i); }
void orc_quat_diff_angle(const float *q0, const float *q1, float *out, int n) { for (int i = 0; i < n; ++i) out[i] = quat_diff_angle(q0 + 4 * i, q1 + 4 * i); }
void orc_slerp(const float *q0, const float *q1, const float *t, float *out, int n) { for (int i = 0; i < n; ++i) slerp(q0 + 4 * i, q1 + 4 * i, t[i], out + 4 * i); }
void orc_calc_heading(const float *q, float *out, int n) { for (int i = 0; i < n; ++i) out[i] = calc_heading(q + 4 * i); }
void orc_rotate_2d_vec(const float *v, const float *angle, float *out, int n) { for (int i = 0; i < n; ++i) rotate_2d_vec(v + 2 * i, angle[i], out + 2 * i); }
void orc_normalize_angle(const float *x, float *out, int n) { for (int i = 0; i < n; ++i) out[i] = normalize_angle(x[i]); }

/* ------------------------------------------------------------------------------------------
 * a3..a6 — PARC/anim/kin_char_model.py
 * ---------------------------------------------------------------------------------------- */
/* dof_to_rot:586 / Joint.dof_to_rot:61 */
static void dof_to_rot1(const OrcChar *c, const float *dof, float *jr) {
    for (int j = 1; j < c->num_bodies; ++j) {
        float *o = jr + 4 * (j - 1);
        const float *d = dof + c->dof_idx[j];
        switch (c->joint_type[j]) {
        case ORC_JOINT_HINGE: axis_angle_to_quat(c->joint_axis[j], d[0], o); break;
        case ORC_JOINT_SPHERICAL: exp_map_to_quat(d, o); break;
        default: o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 1.f; break;
        }
    }
}

/* rot_to_dof:601 / Joint.rot_to_dof:83 */
static void rot_to_dof1(const OrcChar *c, const float *jr, float *dof) {
    for (int i = 0; i < c->dof_size; ++i) dof[i] = 0.f;
    for (int j = 1; j < c->num_bodies; ++j) {
        const float *q = jr + 4 * (j - 1);
        float *d = dof + c->dof_idx[j];
        if (c->joint_type[j] == ORC_JOINT_HINGE) {
            float axis[3], angle;
            quat_to_axis_angle(q, axis, &angle);
            const float *ja = c->joint_axis[j];
            float dot = ja[0] * axis[0] + ja[1] * axis[1] + ja[2] * axis[2];
            if (dot < 0.f) angle = angle * -1.f;
            d[0] = angle;
        } else if (c->joint_type[j] == ORC_JOINT_SPHERICAL) {
            quat_to_exp_map(q, d);
        }
    }
}

/* forward_kinematics:617 */
static void fk1(const OrcChar *c, const float *root_pos, const float *root_rot, const float *jr,
                float *body_pos, float *body_rot_opt) {
    float rot_local[ORC_MAX_BODIES][4];
    float(*br)[4] = rot_local;
    memcpy(body_pos, root_pos, 3 * sizeof(float));
    memcpy(br[0], root_rot, 4 * sizeof(float));
    for (int j = 1; j < c->num_bodies; ++j) {
        int p = c->parent[j];
        float wt[3], cr[4];
        quat_rotate(br[p], c->local_translation[j], wt);
        body_pos[3 * j + 0] = body_pos[3 * p + 0] + wt[0];
        body_pos[3 * j + 1] = body_pos[3 * p + 1] + wt[1];
        body_pos[3 * j + 2] = body_pos[3 * p + 2] + wt[2];
        quat_mul(c->local_rotation[j], jr + 4 * (j - 1), cr);
        quat_mul(br[p], cr, br[j]);
    }
    if (body_rot_opt) memcpy(body_rot_opt, br, (size_t)c->num_bodies * 4 * sizeof(float));
}

/* compute_dof_vel:661 */
static void dof_vel1(const OrcChar *c, const float *jr0, const float *jr1, float dt, float *dv) {
    for (int i = 0; i < c->dof_size; ++i) dv[i] = 0.f;
    for (int j = 1; j < c->num_bodies; ++j) {
        float cj[4], d[4], dn[4], e[3];
        quat_conjugate(jr0 + 4 * (j - 1), cj);
        quat_mul(cj, jr1 + 4 * (j - 1), d);
        quat_normalize(d, dn);
        float *o = dv + c->dof_idx[j];
        if (c->joint_type[j] == ORC_JOINT_HINGE) {
            quat_to_exp_map(dn, e);
            e[0] = e[0] / dt; e[1] = e[1] / dt; e[2] = e[2] / dt;
            const float *ja = c->joint_axis[j];
            o[0] = ja[0] * e[0] + ja[1] * e[1] + ja[2] * e[2];
        } else if (c->joint_type[j] == ORC_JOINT_SPHERICAL) {
            quat_to_exp_map(dn, e);
            o[0] = e[0] / dt; o[1] = e[1] / dt; o[2] = e[2] / dt;
        }
    }
}

void orc_dof_to_rot(const OrcChar *c, const float *dof, float *jr, int n) {
    int J = c->num_bodies - 1;
    for (int i = 0; i < n; ++i) dof_to_rot1(c, dof + (size_t)i * c->dof_size, jr + (size_t)i * J * 4);
}
void orc_rot_to_dof(const OrcChar *c, const float *jr, float *dof, int n) {
    int J = c->num_bodies - 1;
    for (int i = 0; i < n; ++i) rot_to_dof1(c, jr + (size_t)i * J * 4, dof + (size_t)i * c->dof_size);
}
void orc_forward_kinematics(const OrcChar *c, const float *root_pos, const float *root_rot, const float *jr,
                            float *body_pos, float *body_rot, int n) {
    int B = c->num_bodies, J = B - 1;
    for (int i = 0; i < n; ++i)
        fk1(c, root_pos + 3 * (size_t)i, root_rot + 4 * (size_t)i, jr + (size_t)i * J * 4,
            body_pos + (size_t)i * B * 3, body_rot ? body_rot + (size_t)i * B * 4 : NULL);
}
void orc_compute_dof_vel(const OrcChar *c, const float *jr0, const float *jr1, float dt, float *dv, int n) {
    int J = c->num_bodies - 1;
    for (int i = 0; i < n; ++i)
        dof_vel1(c, jr0 + (size_t)i * J * 4, jr1 + (size_t)i * J * 4, dt, dv + (size_t)i * c->dof_size);
}

/* ------------------------------------------------------------------------------------------
 * a7..a9 — PARC/anim/motion_lib.py
 * ---------------------------------------------------------------------------------------- */
OrcMotionLib *orc_mlib_create(const OrcChar *c, int M, const int64_t *num_frames, const int32_t *fps,
                              const int32_t *loop_modes, const double *weights, const float *root_pos,
                              const float *root_rot, const float *joint_rot, const float *contacts) {
    OrcMotionLib *lib = (OrcMotionLib *)calloc(1, sizeof(OrcMotionLib));
    int J = c->num_bodies - 1, B = c->num_bodies, D = c->dof_size;
    int64_t F = 0;
    for (int m = 0; m < M; ++m) F += num_frames[m];
    lib->num_motions = M; lib->num_frames_total = F; lib->num_joints = J; lib->num_bodies = B; lib->dof_size = D;
    lib->motion_weights = (float *)malloc(sizeof(float) * M);
    lib->motion_fps = (float *)malloc(sizeof(float) * M);
    lib->motion_dt = (float *)malloc(sizeof(float) * M);
    lib->motion_lengths = (float *)malloc(sizeof(float) * M);
    lib->motion_root_pos_delta = (float *)malloc(sizeof(float) * 3 * M);
    lib->motion_num_frames = (int64_t *)malloc(sizeof(int64_t) * M);
    lib->motion_start_idx = (int64_t *)malloc(sizeof(int64_t) * M);
    lib->motion_loop_modes = (int32_t *)malloc(sizeof(int32_t) * M);
    lib->frame_root_pos = (float *)malloc(sizeof(float) * 3 * F);
    lib->frame_root_rot = (float *)malloc(sizeof(float) * 4 * F);
    lib->frame_root_vel = (float *)malloc(sizeof(float) * 3 * F);
    lib->frame_root_ang_vel = (float *)malloc(sizeof(float) * 3 * F);
    lib->frame_joint_rot = (float *)malloc(sizeof(float) * 4 * J * F);
    lib->frame_dof_vel = (float *)malloc(sizeof(float) * D * F);
    lib->frame_contacts = (float *)malloc(sizeof(float) * B * F);
    memcpy(lib->frame_root_pos, root_pos, sizeof(float) * 3 * F);
    memcpy(lib->frame_root_rot, root_rot, sizeof(float) * 4 * F);
    memcpy(lib->frame_joint_rot, joint_rot, sizeof(float) * 4 * J * F);
    if (contacts) memcpy(lib->frame_contacts, contacts, sizeof(float) * B * F);
    else memset(lib->frame_contacts, 0, sizeof(float) * B * F); /* motion_lib.py:345-347 */

    float wsum = 0.f;
    for (int m = 0; m < M; ++m) { lib->motion_weights[m] = (float)weights[m]; wsum = wsum + lib->motion_weights[m]; }
    int64_t start = 0;
    for (int m = 0; m < M; ++m) {
        int64_t n = num_frames[m];
        double dt_d = 1.0 / (double)fps[m];                 /* :285 */
        float fpsf = (float)fps[m];
        lib->motion_weights[m] = lib->motion_weights[m] / wsum; /* :372 */
        lib->motion_fps[m] = fpsf;
        lib->motion_dt[m] = (float)dt_d;
        lib->motion_num_frames[m] = n;
        lib->motion_lengths[m] = (float)(1.0 / (double)fps[m] * (double)(n - 1)); /* :305 */
        lib->motion_loop_modes[m] = loop_modes[m];
        lib->motion_start_idx[m] = start;
        const float *rp = root_pos + 3 * start, *rr = root_rot + 4 * start;
        lib->motion_root_pos_delta[3 * m + 0] = rp[3 * (n - 1) + 0] - rp[0]; /* :307-308 */
        lib->motion_root_pos_delta[3 * m + 1] = rp[3 * (n - 1) + 1] - rp[1];
        lib->motion_root_pos_delta[3 * m + 2] = 0.f;
        float *rv = lib->frame_root_vel + 3 * start, *rav = lib->frame_root_ang_vel + 3 * start;
        float *dv = lib->frame_dof_vel + (size_t)D * start;
        const float *jr = joint_rot + (size_t)4 * J * start;
        for (int64_t f = 0; f + 1 < n; ++f) {
            for (int k = 0; k < 3; ++k) rv[3 * f + k] = fpsf * (rp[3 * (f + 1) + k] - rp[3 * f + k]); /* :311 */
            float d[4], e[3];
            quat_diff(rr + 4 * f, rr + 4 * (f + 1), d);                                               /* :315 */
            quat_to_exp_map(d, e);
            for (int k = 0; k < 3; ++k) rav[3 * f + k] = fpsf * e[k];                                  /* :316 */
            dof_vel1(c, jr + (size_t)4 * J * f, jr + (size_t)4 * J * (f + 1), (float)dt_d, dv + (size_t)D * f); /* :319 */
        }
        if (n >= 2) {
            memcpy(rv + 3 * (n - 1), rv + 3 * (n - 2), 3 * sizeof(float));                            /* :312 */
            memcpy(rav + 3 * (n - 1), rav + 3 * (n - 2), 3 * sizeof(float));                          /* :317 */
            memcpy(dv + (size_t)D * (n - 1), dv + (size_t)D * (n - 2), (size_t)D * sizeof(float));    /* kin_char_model.py:656-658 */
        }
        start += n;
    }
    return lib;
}

void orc_mlib_destroy(OrcMotionLib *lib) {
    if (!lib) return;
    free(lib->motion_weights); free(lib->motion_fps); free(lib->motion_dt); free(lib->motion_lengths);
    free(lib->motion_root_pos_delta); free(lib->motion_num_frames); free(lib->motion_start_idx);
    free(lib->motion_loop_modes); free(lib->frame_root_pos); free(lib->frame_root_rot);
    free(lib->frame_root_vel); free(lib->frame_root_ang_vel); free(lib->frame_joint_rot);
    free(lib->frame_dof_vel); free(lib->frame_contacts); free(lib);
}

/* calc_phase:520 + _calc_frame_blend:425 */
static void frame_blend1(const OrcMotionLib *lib, int64_t id, float t, int64_t *i0, int64_t *i1, float *blend) {
    int64_t n = lib->motion_num_frames[id];
    float phase = t / lib->motion_lengths[id];
    if (lib->motion_loop_modes[id] == ORC_LOOP_WRAP) phase = phase - floorf(phase);
    phase = phase < 0.f ? 0.f : (phase > 1.f ? 1.f : phase);
    float pf = phase * (float)(n - 1);
    int64_t f0 = (int64_t)pf; /* .long(): truncation */
    int64_t f1 = f0 + 1 < n - 1 ? f0 + 1 : n - 1;
    *blend = pf - (float)f0;
    *i0 = f0 + lib->motion_start_idx[id];
    *i1 = f1 + lib->motion_start_idx[id];
}

void orc_mlib_calc_frame_blend(const OrcMotionLib *lib, const int64_t *ids, const float *times, int n,
                               int64_t *idx0, int64_t *idx1, float *blend) {
    for (int i = 0; i < n; ++i) frame_blend1(lib, ids[i], times[i], idx0 + i, idx1 + i, blend + i);
}

/* calc_motion_frame:94 (+ _calc_loop_offset:440) — velocities come from frame idx0 un-interpolated */
static void motion_frame1(const OrcMotionLib *lib, int64_t id, float t, float *root_pos, float *root_rot,
                          float *root_vel, float *root_ang_vel, float *joint_rot, float *dof_vel, float *contacts) {
    int64_t i0, i1;
    float b;
    frame_blend1(lib, id, t, &i0, &i1, &b);
    int J = lib->num_joints, B = lib->num_bodies, D = lib->dof_size;
    float a = 1.0f - b;
    for (int k = 0; k < 3; ++k) root_pos[k] = a * lib->frame_root_pos[3 * i0 + k] + b * lib->frame_root_pos[3 * i1 + k];
    slerp(lib->frame_root_rot + 4 * i0, lib->frame_root_rot + 4 * i1, b, root_rot);
    if (root_vel) memcpy(root_vel, lib->frame_root_vel + 3 * i0, 3 * sizeof(float));
    if (root_ang_vel) memcpy(root_ang_vel, lib->frame_root_ang_vel + 3 * i0, 3 * sizeof(float));
    for (int j = 0; j < J; ++j)
        slerp(lib->frame_joint_rot + (size_t)4 * (J * i0 + j), lib->frame_joint_rot + (size_t)4 * (J * i1 + j), b, joint_rot + 4 * j);
    if (dof_vel) memcpy(dof_vel, lib->frame_dof_vel + (size_t)D * i0, (size_t)D * sizeof(float));
    if (lib->motion_loop_modes[id] == ORC_LOOP_WRAP) {
        float ph = floorf(t / lib->motion_lengths[id]);
        for (int k = 0; k < 3; ++k) root_pos[k] = root_pos[k] + ph * lib->motion_root_pos_delta[3 * id + k];
    } else {
        for (int k = 0; k < 3; ++k) root_pos[k] = root_pos[k] + 0.f;
    }
    if (contacts)
        for (int k = 0; k < B; ++k)
            contacts[k] = a * lib->frame_contacts[(size_t)B * i0 + k] + b * lib->frame_contacts[(size_t)B * i1 + k];
}

void orc_mlib_calc_motion_frame(const OrcMotionLib *lib, const int64_t *ids, const float *times, int n,
                                float *root_pos, float *root_rot, float *root_vel, float *root_ang_vel,
                                float *joint_rot, float *dof_vel, float *contacts) {
    int J = lib->num_joints, B = lib->num_bodies, D = lib->dof_size;
    for (int i = 0; i < n; ++i)
        motion_frame1(lib, ids[i], times[i], root_pos + 3 * (size_t)i, root_rot + 4 * (size_t)i,
                      root_vel ? root_vel + 3 * (size_t)i : NULL, root_ang_vel ? root_ang_vel + 3 * (size_t)i : NULL,
                      joint_rot + (size_t)4 * J * i, dof_vel ? dof_vel + (size_t)D * i : NULL,
                      contacts ? contacts + (size_t)B * i : NULL);
}

/* ------------------------------------------------------------------------------------------
 * a11..a13 — terrain_util.py:146-156, geom_util.py:251, mgdm_dm_util.py:128
 * ---------------------------------------------------------------------------------------- */
static inline void grid_index1(const OrcTerrain *t, const float *xy, int64_t *ix, int64_t *iy) {
    float fx = rintf((xy[0] - t->min_point[0]) / t->dxdy[0]); /* torch.round = half-to-even */
    float fy = rintf((xy[1] - t->min_point[1]) / t->dxdy[1]);
    int64_t x = (int64_t)fx, y = (int64_t)fy;
    x = x < 0 ? 0 : (x > t->dims[0] - 1 ? t->dims[0] - 1 : x);
    y = y < 0 ? 0 : (y > t->dims[1] - 1 ? t->dims[1] - 1 : y);
    *ix = x; *iy = y;
}

static inline float hf_val1(const OrcTerrain *t, const float *xy) {
    int64_t x, y;
    grid_index1(t, xy, &x, &y);
    return t->hf[x * t->dims[1] + y];
}

void orc_terrain_grid_index(const OrcTerrain *t, const float *xy, int64_t *idx, int n) {
    for (int i = 0; i < n; ++i) grid_index1(t, xy + 2 * i, idx + 2 * i, idx + 2 * i + 1);
}
void orc_terrain_hf_vals(const OrcTerrain *t, const float *xy, float *out, int n) {
    for (int i = 0; i < n; ++i) out[i] = hf_val1(t, xy + 2 * i);
}

/* geom_util.py:251-272 with torch.linspace's two-sided fp32 formula */
void orc_ray_points_cone(float dx, int num_neg, int num_pos, int rays_neg, int rays_pos, float angle_between, float *out) {
    int dim = num_neg + num_pos + 1;
    float start = (float)(-(double)dx * num_neg), end = (float)((double)dx * num_pos);
    float step = (end - start) / (float)(dim - 1);
    int num_rays = rays_neg + 1 + rays_pos;
    for (int r = 0; r < num_rays; ++r) {
        float ang = (float)(-(double)angle_between * (rays_neg - r));
        for (int i = 0; i < dim; ++i) {
            float x = i < dim / 2 ? start + step * (float)i : end - step * (float)(dim - i - 1);
            float v[2] = {x, 0.f};
            rotate_2d_vec(v, ang, out + 2 * ((size_t)r * dim + i));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a17..a19 standalone pieces
 * ---------------------------------------------------------------------------------------- */
/* mgdm_dm_util.py:335-402.  contact_force / termination_heights ([B] each) are only read when contact_bodies != []
 * (:349-360; termination_heights = terrain height under each body + termination_height, RefCharEnv.update_done :147-152) */
static int32_t compute_done1(const OrcEnvCfg *cfg, int B, float time, const float *root_rot, const float *body_pos,
                             const float *tar_root_rot, const float *tar_body_pos, const float *contact_force,
                             const float *termination_heights) {
    int32_t done = ORC_DONE_NULL;
    if (time >= cfg->episode_length) done = ORC_DONE_TIME;
    if (cfg->enable_early_termination) {
        int failed = 0;
        if (cfg->num_contact_bodies > 0 && contact_force && termination_heights) {
            int fall_contact = 0, fall_height = 0;
            for (int b = 0; b < B; ++b) {
                int is_cb = 0;
                for (int k = 0; k < cfg->num_contact_bodies; ++k) is_cb = is_cb || cfg->contact_body_ids[k] == b;
                if (is_cb) continue; /* masked_contact_buf[:, contact_body_ids] = 0; fall_height[:, contact_body_ids] = False */
                for (int k = 0; k < 3; ++k) if (fabsf(contact_force[3 * b + k]) > 0.1f) fall_contact = 1;
                if (body_pos[3 * b + 2] < termination_heights[b]) fall_height = 1;
            }
            failed = failed || (fall_contact && fall_height);
        }
        if (cfg->pose_termination) {
            int pose_fail = 0;
            for (int b = 1; b < B; ++b) {
                float dist = 0.f;
                for (int k = 0; k < 3; ++k) {
                    float bp = body_pos[3 * b + k] - body_pos[k];
                    float tp = tar_body_pos[3 * b + k] - tar_body_pos[k];
                    float d = tp - bp;
                    dist = dist + d * d;
                }
                float lim = cfg->pose_termination_dist[b - 1];
                if (dist > lim * lim) pose_fail = 1;
            }
            if (cfg->track_root) {
                float dist = 0.f;
                for (int k = 0; k < 3; ++k) {
                    float d = body_pos[k] - tar_body_pos[k];
                    dist = dist + d * d;
                }
                float lim = (float)((double)cfg->root_pos_termination_dist * (double)cfg->root_pos_termination_dist);
                if (dist > lim) pose_fail = 1;
                float err = quat_diff_angle(root_rot, tar_root_rot);
                if (fabsf(err) > cfg->root_rot_termination_angle) pose_fail = 1;
            }
            failed = failed || pose_fail;
        }
        if (!(time > 1e-5f)) failed = 0;
        if (failed) done = ORC_DONE_FAIL;
    }
    return done;
}

void orc_compute_done(const OrcEnvCfg *cfg, int B, const float *time, const float *root_rot, const float *body_pos,
                      const float *tar_root_rot, const float *tar_body_pos, int32_t *done, int n) {
    for (int i = 0; i < n; ++i)
        done[i] = compute_done1(cfg, B, time[i], root_rot + 4 * i, body_pos + (size_t)3 * B * i, tar_root_rot + 4 * i,
                                tar_body_pos + (size_t)3 * B * i, NULL, NULL);
}

/* mgdm_dm_util.py:498-518 */
static inline float contact_reward1(float tar, const float *force, float w) {
    float f = norm3(force);
    if (f > 1.0f) f = 1.0f;
    float r = -(1.0f - tar) * f;
    r = r + tar * f;
    return w * r;
}

void orc_contact_reward(const float *tar, const float *forces, const float *w, float *out, int n, int B) {
    for (int i = 0; i < n; ++i)
        for (int b = 0; b < B; ++b) out[(size_t)i * B + b] = contact_reward1(tar[(size_t)i * B + b], forces + 3 * ((size_t)i * B + b), w[b]);
}

/* ------------------------------------------------------------------------------------------
 * env pipeline
 * ---------------------------------------------------------------------------------------- */
/* dm_env.py:554-565: offset = motion_offsets[mid, tid] - env_offsets[env, 0:2]; pos + offset */
static inline void move_to_motion_terrain(const OrcEnvCfg *cfg, const OrcEnvState *s, int e, float *xy) {
    const float *mo = cfg->motion_offsets + 2 * ((size_t)s->motion_ids[e] * cfg->terrains_per_motion + s->terrain_ids[e]);
    float ox = mo[0] - cfg->env_offsets[3 * e + 0];
    float oy = mo[1] - cfg->env_offsets[3 * e + 1];
    xy[0] = xy[0] + ox;
    xy[1] = xy[1] + oy;
}

/* mgdm_dm_util.py:128-145 via ig_parkour_env.py:518-527 */
static void refresh_rays1(const OrcTerrain *t, const OrcEnvCfg *cfg, OrcEnvState *s, int e) {
    float gx = s->char_root_pos[3 * e + 0] + cfg->env_offsets[3 * e + 0];
    float gy = s->char_root_pos[3 * e + 1] + cfg->env_offsets[3 * e + 1];
    float gz = s->char_root_pos[3 * e + 2] + cfg->env_offsets[3 * e + 2];
    float heading = calc_heading(s->char_root_rot + 4 * e);
    float *out = s->ray_hfs + (size_t)e * cfg->num_rays;
    for (int r = 0; r < cfg->num_rays; ++r) {
        float p[2];
        rotate_2d_vec(cfg->ray_points + 2 * r, heading, p);
        p[0] = p[0] + gx; p[1] = p[1] + gy;
        float h = hf_val1(t, p) - gz;
        h = h < cfg->min_obs_h ? cfg->min_obs_h : h;
        h = h > cfg->max_obs_h ? cfg->max_obs_h : h;
        out[r] = h;
    }
}

void orc_env_refresh_rays(const OrcTerrain *t, const OrcEnvCfg *cfg, OrcEnvState *s, int b, int e_) {
    for (int e = b; e < e_; ++e) refresh_rays1(t, cfg, s, e);
}

/* ig_parkour_env.py:842-965 — obs row = [char 136 | tar S*105 | tar_contacts S*B | char_contacts B | hf R] */
static void compute_obs1(const OrcChar *c, const OrcMotionLib *lib, const OrcEnvCfg *cfg, OrcEnvState *s, int e) {
    int B = c->num_bodies, J = B - 1, D = c->dof_size, K = cfg->num_key, S = cfg->num_tar_steps, R = cfg->num_rays;
    const int rh = cfg->root_height_obs ? 1 : 0;   /* compute_char_obs ig_char_env.py:620-622: [root_h] + obs */
    int char_w = rh + 6 + 3 + 3 + 6 * J + D + 3 * K;
    int tar_w = 3 + 6 + 6 * J + 3 * K;
    const int has_tar = !cfg->no_tar_obs, has_ct = !cfg->no_contact_info;   /* ig_parkour_env.py:927-946: which blocks exist */
    int obs_w = char_w + (has_tar ? S * tar_w : 0) + (has_tar && has_ct ? S * B : 0) + (has_ct ? B : 0) + R;
    float *obs = s->obs + (size_t)e * obs_w;
    float *const row = obs;
    obs += rh;                                      /* the character block as laid out without the root height */
    const float *root_pos = s->char_root_pos + 3 * e, *root_rot = s->char_root_rot + 4 * e;

    float jr[ORC_MAX_BODIES * 4], body_pos[ORC_MAX_BODIES * 3];
    dof_to_rot1(c, s->char_dof_pos + (size_t)D * e, jr);           /* :865 */
    fk1(c, root_pos, root_rot, jr, body_pos, NULL);                 /* :868 */

    /* compute_char_obs ig_char_env.py:582-627; global_obs: the raw root rotation / velocities / key offsets */
    const int gl = cfg->global_obs;
    if (rh) row[0] = root_pos[2];
    float hinv[4], lr[4];
    calc_heading_quat_inv(root_rot, hinv);
    if (gl) {
        quat_to_tan_norm(root_rot, obs + 0);
        memcpy(obs + 6, s->char_root_vel + 3 * e, 3 * sizeof(float));
        memcpy(obs + 9, s->char_root_ang_vel + 3 * e, 3 * sizeof(float));
    } else {
        quat_mul(hinv, root_rot, lr);
        quat_to_tan_norm(lr, obs + 0);
        quat_rotate(hinv, s->char_root_vel + 3 * e, obs + 6);
        quat_rotate(hinv, s->char_root_ang_vel + 3 * e, obs + 9);
    }
    for (int j = 0; j < J; ++j) quat_to_tan_norm(jr + 4 * j, obs + 12 + 6 * j);
    memcpy(obs + 12 + 6 * J, s->char_dof_vel + (size_t)D * e, (size_t)D * sizeof(float));
    for (int k = 0; k < K; ++k) {
        float rel[3];
        int b = cfg->key_body_ids[k];
        for (int a = 0; a < 3; ++a) rel[a] = body_pos[3 * b + a] - root_pos[a];
        if (gl) memcpy(obs + 12 + 6 * J + D + 3 * k, rel, sizeof(rel));
        else quat_rotate(hinv, rel, obs + 12 + 6 * J + D + 3 * k);
    }

    /* DeepMimicEnv.compute_tar_obs dm_env.py:594-626 + fetch_tar_obs_data mgdm_dm_util.py:221 + compute_tar_obs :405 */
    float mt = s->time_buf[e] + s->time_offsets[e]; /* _get_motion_times:547 */
    float *tar = row + char_w;
    float *tarc = tar + (has_tar ? S * tar_w : 0);
    float tarc_scratch[ORC_MAX_BODIES];
    for (int si = 0; si < S && has_tar; ++si) {
        float tstep = (float)cfg->timestep_d * (float)cfg->tar_obs_steps[si]; /* timestep * tar_obs_steps (f32 tensor) */
        float t = mt + tstep;
        float trp[3], trr[4], tjr[ORC_MAX_BODIES * 4], tbp[ORC_MAX_BODIES * 3];
        motion_frame1(lib, s->motion_ids[e], t, trp, trr, NULL, NULL, tjr, NULL, has_ct ? tarc + (size_t)si * B : tarc_scratch);
        move_to_motion_terrain(cfg, s, e, trp);
        fk1(c, trp, trr, tjr, tbp, NULL);
        float *o = tar + (size_t)si * tar_w;
        float rpo[3], rpl[3], lrr[4];
        for (int a = 0; a < 3; ++a) rpo[a] = trp[a] - root_pos[a];
        if (gl) { /* mgdm_dm_util.py:417: nothing is turned into the heading frame, and the key offsets stay relative to the target root */
            o[0] = rpo[0]; o[1] = rpo[1]; o[2] = rpo[2];
            quat_to_tan_norm(trr, o + 3);
        } else {
            quat_rotate(hinv, rpo, rpl);
            o[0] = rpl[0]; o[1] = rpl[1]; o[2] = rpl[2];
            quat_mul(hinv, trr, lrr);
            quat_to_tan_norm(lrr, o + 3);
        }
        for (int j = 0; j < J; ++j) quat_to_tan_norm(tjr + 4 * j, o + 9 + 6 * j);
        for (int k = 0; k < K; ++k) {
            int b = cfg->key_body_ids[k];
            float rel[3], rl[3];
            for (int a = 0; a < 3; ++a) rel[a] = tbp[3 * b + a] - trp[a];
            if (gl) {
                for (int a = 0; a < 3; ++a) o[9 + 6 * J + 3 * k + a] = rel[a];
            } else {
                quat_rotate(hinv, rel, rl);
                for (int a = 0; a < 3; ++a) o[9 + 6 * J + 3 * k + a] = rl[a] + rpl[a];
            }
        }
    }
    /* char contacts ig_parkour_env.py:655-662 */
    float *cc = tarc + (has_tar && has_ct ? (size_t)S * B : 0);
    for (int b = 0; b < B && has_ct; ++b) cc[b] = norm3(s->contact_forces + 3 * ((size_t)e * B + b)) > 1e-5f ? 1.f : 0.f;
    memcpy(cc + (has_ct ? B : 0), s->ray_hfs + (size_t)e * R, (size_t)R * sizeof(float));
}

void orc_env_compute_obs(const OrcChar *c, const OrcMotionLib *lib, const OrcEnvCfg *cfg, OrcEnvState *s,
                         const int64_t *env_ids, int k) {
    for (int i = 0; i < k; ++i) compute_obs1(c, lib, cfg, s, (int)env_ids[i]);
}

/* mgdm_dm_util.py:246-267 */
static void convert_to_local(float *root_rot, float *root_vel, float *root_ang_vel, float *key_pos, int K) {
    float hinv[4], t4[4], t3[3];
    calc_heading_quat_inv(root_rot, hinv);
    quat_mul(hinv, root_rot, t4); memcpy(root_rot, t4, sizeof(t4));
    quat_rotate(hinv, root_vel, t3); memcpy(root_vel, t3, sizeof(t3));
    quat_rotate(hinv, root_ang_vel, t3); memcpy(root_ang_vel, t3, sizeof(t3));
    for (int k = 0; k < K; ++k) { quat_rotate(hinv, key_pos + 3 * k, t3); memcpy(key_pos + 3 * k, t3, sizeof(t3)); }
}

/* ig_parkour_env.py:984-1096 + mgdm_dm_util.py:270-333,498-518,521-553 */
static void update_reward1(const OrcChar *c, const OrcEnvCfg *cfg, OrcEnvState *s, int e) {
    int B = c->num_bodies, J = B - 1, D = c->dof_size, K = cfg->num_key, N = cfg->num_envs;
    float jr[ORC_MAX_BODIES * 4];
    dof_to_rot1(c, s->char_dof_pos + (size_t)D * e, jr);
    const float *tjr = s->ref_joint_rot + (size_t)4 * J * e;
    float pose_err = 0.f;
    for (int j = 0; j < J; ++j) {
        float d = quat_diff_angle(jr + 4 * j, tjr + 4 * j);
        pose_err = pose_err + cfg->joint_err_w[j] * d * d;
    }
    float vel_err = 0.f;
    for (int d = 0; d < D; ++d) {
        float v = s->ref_dof_vel[(size_t)D * e + d] - s->char_dof_vel[(size_t)D * e + d];
        vel_err = vel_err + cfg->dof_err_w[d] * v * v;
    }
    float root_pos[3], tar_root_pos[3], rpd[3];
    for (int a = 0; a < 3; ++a) {
        root_pos[a] = s->char_root_pos[3 * e + a];
        tar_root_pos[a] = s->ref_root_pos[3 * e + a];
        rpd[a] = tar_root_pos[a] - root_pos[a];
    }
    if (!cfg->track_root) { rpd[0] = 0.f; rpd[1] = 0.f; }
    if (!cfg->track_root_h) rpd[2] = 0.f;
    float root_pos_err = rpd[0] * rpd[0] + rpd[1] * rpd[1] + rpd[2] * rpd[2];

    float key[ORC_MAX_KEY * 3], tkey[ORC_MAX_KEY * 3];
    for (int k = 0; k < K; ++k) {
        int b = cfg->key_body_ids[k];
        for (int a = 0; a < 3; ++a) {
            key[3 * k + a] = s->char_body_pos[3 * ((size_t)e * B + b) + a] - root_pos[a];   /* simulator body pos :987 */
            tkey[3 * k + a] = s->ref_body_pos[3 * ((size_t)e * B + b) + a] - tar_root_pos[a];
        }
    }
    float rr[4], rv[3], rav[3], trr[4], trv[3], trav[3];
    memcpy(rr, s->char_root_rot + 4 * e, sizeof(rr)); memcpy(trr, s->ref_root_rot + 4 * e, sizeof(trr));
    memcpy(rv, s->char_root_vel + 3 * e, sizeof(rv)); memcpy(trv, s->ref_root_vel + 3 * e, sizeof(trv));
    memcpy(rav, s->char_root_ang_vel + 3 * e, sizeof(rav)); memcpy(trav, s->ref_root_ang_vel + 3 * e, sizeof(trav));
    if (!cfg->track_root) {
        convert_to_local(rr, rv, rav, key, K);
        convert_to_local(trr, trv, trav, tkey, K);
    }
    float root_rot_err = quat_diff_angle(rr, trr);
    root_rot_err = root_rot_err * root_rot_err;
    float root_vel_err = 0.f, root_ang_vel_err = 0.f;
    for (int a = 0; a < 3; ++a) { float d = trv[a] - rv[a]; root_vel_err = root_vel_err + d * d; }
    for (int a = 0; a < 3; ++a) { float d = trav[a] - rav[a]; root_ang_vel_err = root_ang_vel_err + d * d; }
    float key_pos_err = 0.f;
    for (int k = 0; k < K; ++k) {
        float ke = 0.f;
        for (int a = 0; a < 3; ++a) { float d = tkey[3 * k + a] - key[3 * k + a]; ke = ke + d * d; }
        key_pos_err = key_pos_err + ke;
    }
    float pose_r = expf(-0.25f * pose_err);
    float vel_r = expf(-0.01f * vel_err);
    float root_pose_r = expf(-5.0f * (root_pos_err + 0.1f * root_rot_err));
    float root_vel_r = expf(-1.0f * (root_vel_err + 0.1f * root_ang_vel_err));
    float key_pos_r = expf(-10.0f * key_pos_err);

    float r = cfg->pose_w * pose_r + cfg->vel_w * vel_r + cfg->root_pos_w * root_pose_r + cfg->root_vel_w * root_vel_r +
              cfg->key_pos_w * key_pos_r;
    float csum = 0.f;
    for (int b = 0; b < B; ++b)
        csum = csum + contact_reward1(s->ref_contacts[(size_t)B * e + b], s->contact_forces + 3 * ((size_t)e * B + b), cfg->contact_weights[b]);
    float contact_penalty = cfg->no_contact_info ? 0.f : csum / (float)B;   /* ig_parkour_env.py:1032: only with use_contact_info */
    r = r + contact_penalty;
    s->reward[e] = r;
    float *rt = s->reward_terms;
    rt[0 * (size_t)N + e] = pose_r; rt[1 * (size_t)N + e] = vel_r; rt[2 * (size_t)N + e] = root_pose_r;
    rt[3 * (size_t)N + e] = root_vel_r; rt[4 * (size_t)N + e] = key_pos_r; rt[5 * (size_t)N + e] = contact_penalty;
    rt[6 * (size_t)N + e] = r;

    if (s->tracking_error) { /* mgdm_dm_util.py:521-553 with FK of both skeletons (ig_parkour_env.py:1060-1088) */
        float cbp[ORC_MAX_BODIES * 3], cbr[ORC_MAX_BODIES * 4], rbp[ORC_MAX_BODIES * 3], rbr[ORC_MAX_BODIES * 4];
        fk1(c, root_pos, s->char_root_rot + 4 * e, jr, cbp, cbr);
        fk1(c, tar_root_pos, s->ref_root_rot + 4 * e, tjr, rbp, rbr);
        float pe = 0.f, bpe = 0.f;
        for (int b = 0; b < B; ++b) {
            pe = pe + fabsf(quat_diff_angle(cbr + 4 * b, rbr + 4 * b));
            float d[3];
            for (int a = 0; a < 3; ++a) d[a] = (rbp[3 * b + a] - tar_root_pos[a]) - (cbp[3 * b + a] - root_pos[a]);
            bpe = bpe + norm3(d);
        }
        float rd[3];
        for (int a = 0; a < 3; ++a) rd[a] = tar_root_pos[a] - root_pos[a];
        float dve = 0.f;
        for (int d = 0; d < D; ++d) dve = dve + fabsf(s->ref_dof_vel[(size_t)D * e + d] - s->char_dof_vel[(size_t)D * e + d]);
        float rve = 0.f, rave = 0.f;
        for (int a = 0; a < 3; ++a) {
            rve = rve + fabsf(s->ref_root_vel[3 * e + a] - s->char_root_vel[3 * e + a]);
            rave = rave + fabsf(s->ref_root_ang_vel[3 * e + a] - s->char_root_ang_vel[3 * e + a]);
        }
        float *te = s->tracking_error + 7 * (size_t)e;
        te[0] = norm3(rd);
        te[1] = fabsf(quat_diff_angle(s->char_root_rot + 4 * e, s->ref_root_rot + 4 * e));
        te[2] = bpe / (float)B;
        te[3] = pe / (float)B;
        te[4] = dve / (float)D;
        te[5] = rve / 3.f;
        te[6] = rave / 3.f;
    }
}

/* ig_env.py:368-377 order for one env: rays -> time -> ref motion -> obs -> reward -> done(pre-curriculum) */
void orc_env_post_physics_step(const OrcChar *c, const OrcMotionLib *lib, const OrcTerrain *t, const OrcEnvCfg *cfg,
                               OrcEnvState *s, int env_begin, int env_end) {
    int B = c->num_bodies, J = B - 1, D = c->dof_size;
    for (int e = env_begin; e < env_end; ++e) {
        refresh_rays1(t, cfg, s, e);
        /* _update_time ig_env.py:391-394 */
        s->timestep_buf[e] += 1;
        s->time_buf[e] = (float)cfg->timestep_d * (float)s->timestep_buf[e];
        /* _update_ref_motion dm_env.py:523-545 */
        float mt = s->time_buf[e] + s->time_offsets[e];
        float *rp = s->ref_root_pos + 3 * e;
        motion_frame1(lib, s->motion_ids[e], mt, rp, s->ref_root_rot + 4 * e, s->ref_root_vel + 3 * e,
                      s->ref_root_ang_vel + 3 * e, s->ref_joint_rot + (size_t)4 * J * e, s->ref_dof_vel + (size_t)D * e,
                      s->ref_contacts + (size_t)B * e);
        move_to_motion_terrain(cfg, s, e, rp);
        fk1(c, rp, s->ref_root_rot + 4 * e, s->ref_joint_rot + (size_t)4 * J * e, s->ref_body_pos + (size_t)3 * B * e, NULL);
        rot_to_dof1(c, s->ref_joint_rot + (size_t)4 * J * e, s->ref_dof_pos + (size_t)D * e);
        compute_obs1(c, lib, cfg, s, e);
        update_reward1(c, cfg, s, e);
        /* RefCharEnv.update_done mgdm_dm_util.py:147-152: the terrain height under every body (global xy = env-local + env offset) */
        float term_h[16];
        if (cfg->num_contact_bodies > 0)
            for (int b = 0; b < B; ++b) {
                const float *bp = s->char_body_pos + 3 * ((size_t)B * e + b);
                float gxy[2] = {bp[0] + cfg->env_offsets[3 * e + 0], bp[1] + cfg->env_offsets[3 * e + 1]};
                term_h[b] = hf_val1(t, gxy) + cfg->termination_height;
            }
        s->done[e] = compute_done1(cfg, B, s->time_buf[e], s->char_root_rot + 4 * e, s->char_body_pos + (size_t)3 * B * e,
                                   s->ref_root_rot + 4 * e, s->ref_body_pos + (size_t)3 * B * e,
                                   s->contact_forces + (size_t)3 * B * e, term_h);
    }
}

/* dm_env.py:636-665: motion end, sequential EMA in env order, motion_end => FAIL */
void orc_env_update_curriculum(const OrcMotionLib *lib, const OrcEnvCfg *cfg, OrcEnvState *s) {
    float w = cfg->ema_weight;
    float keep = (float)(1.0 - (double)w);
    for (int e = 0; e < cfg->num_envs; ++e) {
        int64_t mid = s->motion_ids[e];
        float mt = s->time_buf[e] + s->time_offsets[e];
        int motion_end = (mt >= lib->motion_lengths[mid]) && (lib->motion_loop_modes[mid] != ORC_LOOP_WRAP);
        if (s->done[e] != ORC_DONE_NULL || motion_end) {
            if (s->done[e] == ORC_DONE_FAIL) s->fail_rates[mid] = s->fail_rates[mid] * keep + w;
            else s->fail_rates[mid] = s->fail_rates[mid] * keep;
        }
        if (motion_end) s->done[e] = ORC_DONE_FAIL;
    }
}

/* dm_env.py:567-592 (_reset_ref_motion:473, _char_state_init_from_ref mgdm_dm_util.py:89, add_noise :102) */
void orc_env_reset_with(const OrcChar *c, const OrcMotionLib *lib, const OrcTerrain *t, const OrcEnvCfg *cfg,
                        OrcEnvState *s, const int64_t *env_ids, int k, const int64_t *motion_ids,
                        const int64_t *terrain_ids, const float *t0, const float *xy_noise) {
    (void)t;
    int B = c->num_bodies, J = B - 1, D = c->dof_size;
    for (int i = 0; i < k; ++i) {
        int e = (int)env_ids[i];
        s->motion_ids[e] = motion_ids[i];
        s->terrain_ids[e] = terrain_ids[i];
        s->time_offsets[e] = t0[i];
        float *rp = s->ref_root_pos + 3 * e;
        motion_frame1(lib, motion_ids[i], t0[i], rp, s->ref_root_rot + 4 * e, s->ref_root_vel + 3 * e,
                      s->ref_root_ang_vel + 3 * e, s->ref_joint_rot + (size_t)4 * J * e, s->ref_dof_vel + (size_t)D * e,
                      s->ref_contacts + (size_t)B * e);
        move_to_motion_terrain(cfg, s, e, rp);
        rot_to_dof1(c, s->ref_joint_rot + (size_t)4 * J * e, s->ref_dof_pos + (size_t)D * e);
        memcpy(s->char_root_pos + 3 * e, rp, 3 * sizeof(float));
        memcpy(s->char_root_rot + 4 * e, s->ref_root_rot + 4 * e, 4 * sizeof(float));
        memcpy(s->char_root_vel + 3 * e, s->ref_root_vel + 3 * e, 3 * sizeof(float));
        memcpy(s->char_root_ang_vel + 3 * e, s->ref_root_ang_vel + 3 * e, 3 * sizeof(float));
        memcpy(s->char_dof_pos + (size_t)D * e, s->ref_dof_pos + (size_t)D * e, (size_t)D * sizeof(float));
        memcpy(s->char_dof_vel + (size_t)D * e, s->ref_dof_vel + (size_t)D * e, (size_t)D * sizeof(float));
        s->char_root_pos[3 * e + 0] = s->char_root_pos[3 * e + 0] + xy_noise[2 * i + 0];
        s->char_root_pos[3 * e + 1] = s->char_root_pos[3 * e + 1] + xy_noise[2 * i + 1];
        s->timestep_buf[e] = 0;
        s->time_buf[e] = 0.f;
        s->done[e] = ORC_DONE_NULL;
    }
}
