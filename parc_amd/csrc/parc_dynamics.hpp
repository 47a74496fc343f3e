// parc_dynamics.hpp — reduced-coordinate rigid-body dynamics of the articulated humanoid (replaces the
// Isaac Gym / PhysX calls of the reference: ig_env.py:359-366 `gym.simulate` x sim_steps with `substeps`,
// PD position drives ig_char_env.py:121-136,488-497, terrain contact ig_util.py:8-24).
//
// PhysX is a closed binary that is not in the container: THIS IS A RE-AUTHORED SIMULATOR, parity with PhysX is
// unpinned.  It is validated by invariants (momentum in free flight, energy decay, resting contact, PD step
// response) and by CPU-vs-GPU equality of this same code.
//
// Formulation (one env per THREAD: every lane of a wave advances a different character, so the serial tree
// recursion costs one instruction per 64 envs):
//   * Featherstone articulated-body algorithm in a common frame: world axes, reference point O = root origin,
//     spatial vectors ordered [angular; linear].  Joint types: SPHERICAL (3 dofs = relative angular velocity in
//     the child frame, position = exp map, as Isaac Gym reports merged hinge triples), HINGE, FIXED.
//   * PD drives are integrated implicitly: tau = kp*err - (kd + dt*kp)*qd explicit part, and dt*kd + dt^2*kp +
//     armature added to the joint-space diagonal D (backward Euler on the drive) — stable for the MJCF gains
//     (kp up to 1000, kd up to 100) at dt = 1/120 s.
//   * Contact: collision spheres / box corners against the heightfield's cell columns (cell (i,j) is an axis
//     aligned column topped at hf[i,j], the reference's blocky mesh terrain_util.py:1099-1184), spring-damper
//     normal force + regularised Coulomb friction, linearised in the body acceleration and folded into the
//     articulated inertia (implicit), so stiff contacts stay stable at this step size.
//   * Semi-implicit Euler on velocities, quaternion integration of the root and the spherical joints.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/parc_env.h"

#if defined(__HIPCC__)
#define PARC_HD __host__ __device__ __forceinline__
#define PARC_UNROLL _Pragma("unroll")
#else
#define PARC_HD inline
#define PARC_UNROLL
#endif

// Device-side reciprocal / square root / sin / cos of the dynamics use the hardware approximations (v_rcp_f32, v_sqrt_f32,
// v_sin_f32: ~1 ulp, resp. ~1e-6 absolute for the half-angles that occur here) instead of the correctly rounded OCML
// sequences: nothing here has a bit-exact reference and the integrator's own truncation error is orders larger.  The
// host build (tests) uses libm.
#if defined(__HIP_DEVICE_COMPILE__)
#define DYN_RCP(x) __builtin_amdgcn_rcpf(x)
#define DYN_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define DYN_RINT(x) __builtin_rintf(x)
#define DYN_SIN(x) __sinf(x)
#define DYN_COS(x) __cosf(x)
#else
#define DYN_RCP(x) (1.f / (x))
#define DYN_SQRT(x) sqrtf(x)
#define DYN_RINT(x) rintf(x)
#define DYN_SIN(x) sinf(x)
#define DYN_COS(x) cosf(x)
#endif

// The dynamics has no bit-exact reference (parity with PhysX is unpinned, the CPU build is compared by tolerance), so the
// device build may fuse multiply-adds here even though the rest of the library is built with -ffp-contract=off.
#if defined(__HIPCC__)
#pragma clang fp contract(fast)
#endif

namespace parcdyn {

#define DYN_MAXB 16
#define DYN_MAXD 40
#define DYN_MAXC 48
#define DYN_MAXS 24  // collision segments (capsule axes, sole edges of boxes)
#define DYN_PATCH 9 // local height patch (cells) around the root

enum { DJ_ROOT = 0, DJ_HINGE = 1, DJ_SPHERICAL = 2, DJ_FIXED = 3 };

struct DynModel {
    int B, D;
    int parent[DYN_MAXB], jtype[DYN_MAXB], dof_idx[DYN_MAXB];
    float lt[DYN_MAXB][3];       // joint origin in the parent frame
    float lr[DYN_MAXB][4];       // fixed local rotation (xyzw)
    float axis[DYN_MAXB][3];     // hinge axis (joint frame)
    float mass[DYN_MAXB];
    float com[DYN_MAXB][3];      // body frame
    float inertia[DYN_MAXB][6];  // about the COM, body frame: xx yy zz xy xz yz
    float kp[DYN_MAXD], kd[DYN_MAXD], arm[DYN_MAXD], eff[DYN_MAXD], lo[DYN_MAXD], hi[DYN_MAXD];
    float act_lo[DYN_MAXD], act_hi[DYN_MAXD];
    int ncol;
    int col_body[DYN_MAXC];
    float col_pos[DYN_MAXC][3];  // body frame
    float col_r[DYN_MAXC];
    // collision SEGMENTS: a capsule's axis (radius seg_r) or one of the two long sole edges of a box (a thin capsule, radius 1 cm).  The points above test a
    // capsule's two end spheres / a box's corners; a segment adds the contact of the shaft (edge) with a column's top EDGE where it
    // crosses a grid line between two columns of different height (segment_edge_point): a shin lying across a platform edge, a sole
    // that comes down on the lip of a step between its corners.  Sorted by body like the points.
    int nseg;
    int seg_body[DYN_MAXS];
    float seg_a[DYN_MAXS][3], seg_b[DYN_MAXS][3];  // end points, body frame
    float seg_r[DYN_MAXS];
    float gravity_z, dt;         // dt of one solver substep
    int nsub;                    // substeps per control step (sim_steps * substeps)
    float kn, dn, dtang, mu;     // contact stiffness / normal damping / tangential damping / friction
    float pen_cap;               // penetration beyond which the contact spring saturates (PhysX max_depenetration_velocity, see fill_dyn_model)
    // CONTACT MANIFOLD (round 4).  Contacts are DISCOVERED (candidate spheres against the cell columns: the expensive, divergent part)
    // in substeps 0, man_period, 2 man_period, ... of a control step and kept as PLANES (body-frame point, world normal, offset, weight);
    // every substep re-evaluates penetration and velocity along the cached planes (contact_apply on pen > 0).  Discovery also keeps
    // SPECULATIVE planes: candidates that are not touching yet but could within the substeps that follow (pen > -margin, margin =
    // min(spec_m0 + spec_tv * approach speed along the normal, spec_max)), so a foot coming down is caught by the plane it is about to hit.
    // PhysX does the same in spirit: shapes closer than contact_offset (0.02 m, dm_env_default.yaml sim.physx) generate contacts before
    // they touch, and its persistent contact manifold is re-used while the relative pose has moved little.  man_period = 1 is the
    // round-3 behaviour (discovery in every substep; the speculative planes then never carry a force).
    int man_period;
    float spec_m0, spec_tv, spec_max;
    float lim_k, lim_d;          // joint-limit penalty
    float max_ang_vel, ang_damping;
    float total_mass;
    int ctrl;                    // control mode (PARC_CTRL_*, ig_char_env.py:21-26): 0 = pd, the implicit drive towards the action's pose; the others
                                 // turn the action into a feed-forward joint torque that is constant over the control step (ctrl_ff / drive_ff below)
    int truncated;               // collision points / segments / mass parts that did not fit the fixed tables (DYN_MAXC / DYN_MAXS / 64): the model
                                 // would depend on the geom order -- parc_env_create refuses such a model
    float ext_acc[2];            // TEST HOOK of the host build (oracle/dyn_oracle.cpp): uniform horizontal acceleration, i.e. a tilted gravity
                                 // vector (a slope without tilting the heightfield).  Always 0 in the library: fill_dyn_model zeroes it and
                                 // only this header's reference statement reads it.
};

struct DynTerrain {
    const float *hf;
    int X, Y;
    float min_x, min_y, dx, dy;
};

// ---------------------------------------------------------------- small vector helpers
struct v3 { float x, y, z; };
PARC_HD v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PARC_HD v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PARC_HD v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PARC_HD v3 operator*(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
PARC_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PARC_HD v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
struct q4 { float x, y, z, w; };
PARC_HD q4 qmul(q4 a, q4 b) {
    q4 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
PARC_HD q4 qconj(q4 a) { q4 r; r.x = -a.x; r.y = -a.y; r.z = -a.z; r.w = a.w; return r; }
PARC_HD q4 qnormalize(q4 a) {
    float n = DYN_SQRT(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
    n = n > 1e-12f ? DYN_RCP(n) : 0.f;
    q4 r; r.x = a.x * n; r.y = a.y * n; r.z = a.z * n; r.w = a.w * n;
    if (n == 0.f) r.w = 1.f;
    return r;
}
// rotation-vector -> quaternion
PARC_HD q4 qexp(v3 v) {
    float a = DYN_SQRT(dot(v, v));
    q4 r;
    if (a < 1e-6f) { r.x = 0.5f * v.x; r.y = 0.5f * v.y; r.z = 0.5f * v.z; r.w = 1.f; return qnormalize(r); }
    float s = DYN_SIN(0.5f * a) * DYN_RCP(a);
    r.x = s * v.x; r.y = s * v.y; r.z = s * v.z; r.w = DYN_COS(0.5f * a);
    return r;
}
// atan2(y, x) for y >= 0, x >= 0 (not both 0).  Device: arctan(t) / t = 1 + sum a_2k t^2k on [0, 1] (Abramowitz & Stegun 4.4.49, 8 terms;
// 1e-7 absolute in fp32, checked on the CPU) -- 15 instructions instead of OCML's atan2f with its IEEE division and special-case fix-ups;
// a spherical joint's elimination calls it twice per substep (PD error, joint limits).  The host build uses libm.
PARC_HD float dyn_atan2_pos(float y, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float mx = x > y ? x : y, mn = x > y ? y : x;
    const float t = mn * DYN_RCP(mx), t2 = t * t;
    float p = 0.0028662257f;
    p = p * t2 - 0.0161657367f; p = p * t2 + 0.0429096138f; p = p * t2 - 0.0752896400f; p = p * t2 + 0.1065626393f;
    p = p * t2 - 0.1420889944f; p = p * t2 + 0.1999355085f; p = p * t2 - 0.3333314528f; p = p * t2 + 1.f;
    p *= t;
    return y > x ? 1.57079632679489662f - p : p;
#else
    return atan2f(y, x);
#endif
}
// quaternion -> rotation vector with angle in [0, pi] (the reference's quat_to_exp_map convention)
PARC_HD v3 qlog(q4 q) {
    if (q.w < 0.f) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    float l = DYN_SQRT(q.x * q.x + q.y * q.y + q.z * q.z);
    if (l < 1e-6f) return mk(2.f * q.x, 2.f * q.y, 2.f * q.z);
    float a = 2.f * dyn_atan2_pos(l, q.w) * DYN_RCP(l);
    return mk(a * q.x, a * q.y, a * q.z);
}
struct m3 { float m[3][3]; };
PARC_HD m3 qmat(q4 q) {
    m3 R;
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    R.m[0][0] = 1.f - 2.f * (yy + zz); R.m[0][1] = 2.f * (xy - wz); R.m[0][2] = 2.f * (xz + wy);
    R.m[1][0] = 2.f * (xy + wz); R.m[1][1] = 1.f - 2.f * (xx + zz); R.m[1][2] = 2.f * (yz - wx);
    R.m[2][0] = 2.f * (xz - wy); R.m[2][1] = 2.f * (yz + wx); R.m[2][2] = 1.f - 2.f * (xx + yy);
    return R;
}
PARC_HD v3 mulv(const m3 &R, v3 v) {
    return mk(R.m[0][0] * v.x + R.m[0][1] * v.y + R.m[0][2] * v.z, R.m[1][0] * v.x + R.m[1][1] * v.y + R.m[1][2] * v.z,
              R.m[2][0] * v.x + R.m[2][1] * v.y + R.m[2][2] * v.z);
}
PARC_HD v3 mulTv(const m3 &R, v3 v) {
    return mk(R.m[0][0] * v.x + R.m[1][0] * v.y + R.m[2][0] * v.z, R.m[0][1] * v.x + R.m[1][1] * v.y + R.m[2][1] * v.z,
              R.m[0][2] * v.x + R.m[1][2] * v.y + R.m[2][2] * v.z);
}

// ---------------------------------------------------------------- 6-D spatial algebra, [angular; linear]
struct s6 { float a[6]; };
PARC_HD s6 s6zero() { s6 r; for (int i = 0; i < 6; ++i) r.a[i] = 0.f; return r; }
PARC_HD s6 s6mk(v3 w, v3 v) { s6 r; r.a[0] = w.x; r.a[1] = w.y; r.a[2] = w.z; r.a[3] = v.x; r.a[4] = v.y; r.a[5] = v.z; return r; }
PARC_HD v3 s6ang(const s6 &s) { return mk(s.a[0], s.a[1], s.a[2]); }
PARC_HD v3 s6lin(const s6 &s) { return mk(s.a[3], s.a[4], s.a[5]); }
PARC_HD s6 operator+(const s6 &a, const s6 &b) { s6 r; for (int i = 0; i < 6; ++i) r.a[i] = a.a[i] + b.a[i]; return r; }
// motion cross  v x m
PARC_HD s6 crm(const s6 &v, const s6 &m) {
    v3 w = s6ang(v), vl = s6lin(v), mw = s6ang(m), ml = s6lin(m);
    return s6mk(cross(w, mw), cross(w, ml) + cross(vl, mw));
}
// force cross  v x* f
PARC_HD s6 crf(const s6 &v, const s6 &f) {
    v3 w = s6ang(v), vl = s6lin(v), n = s6ang(f), fl = s6lin(f);
    return s6mk(cross(w, n) + cross(vl, fl), cross(w, fl));
}
// symmetric 6x6, packed upper triangle row-major (21 entries)
struct sym6 { float s[21]; };
PARC_HD int sidx(int i, int j) { if (i > j) { int t = i; i = j; j = t; } return i * 6 - (i * (i - 1)) / 2 + (j - i); }
PARC_HD float sget(const sym6 &A, int i, int j) { return A.s[sidx(i, j)]; }
PARC_HD s6 symmul(const sym6 &A, const s6 &x) {
    s6 r;
    PARC_UNROLL
    for (int i = 0; i < 6; ++i) {
        float acc = 0.f;
        PARC_UNROLL
        for (int j = 0; j < 6; ++j) acc += sget(A, i, j) * x.a[j];
        r.a[i] = acc;
    }
    return r;
}
// A += k * w w^T
PARC_HD void symrank1(sym6 &A, float k, const s6 &w) {
    PARC_UNROLL
    for (int i = 0; i < 6; ++i) {
        PARC_UNROLL
        for (int j = i; j < 6; ++j) A.s[sidx(i, j)] += k * w.a[i] * w.a[j];
    }
}
// spatial inertia about O of a point-like mass m at c plus rotational inertia Ic (world axes): adds into A
PARC_HD void add_inertia(sym6 &A, float m, v3 c, const float Ic[6] /* xx yy zz xy xz yz or null */) {
    float cc = dot(c, c);
    A.s[sidx(0, 0)] += m * (cc - c.x * c.x); A.s[sidx(1, 1)] += m * (cc - c.y * c.y); A.s[sidx(2, 2)] += m * (cc - c.z * c.z);
    A.s[sidx(0, 1)] += -m * c.x * c.y; A.s[sidx(0, 2)] += -m * c.x * c.z; A.s[sidx(1, 2)] += -m * c.y * c.z;
    if (Ic) {
        A.s[sidx(0, 0)] += Ic[0]; A.s[sidx(1, 1)] += Ic[1]; A.s[sidx(2, 2)] += Ic[2];
        A.s[sidx(0, 1)] += Ic[3]; A.s[sidx(0, 2)] += Ic[4]; A.s[sidx(1, 2)] += Ic[5];
    }
    // upper-right block m [c]x :  [[0,-cz,cy],[cz,0,-cx],[-cy,cx,0]]
    A.s[sidx(0, 4)] += -m * c.z; A.s[sidx(0, 5)] += m * c.y;
    A.s[sidx(1, 3)] += m * c.z;  A.s[sidx(1, 5)] += -m * c.x;
    A.s[sidx(2, 3)] += -m * c.y; A.s[sidx(2, 4)] += m * c.x;
    A.s[sidx(3, 3)] += m; A.s[sidx(4, 4)] += m; A.s[sidx(5, 5)] += m;
}

// ---------------------------------------------------------------- terrain: local height patch
struct Patch {
    float h[DYN_PATCH * DYN_PATCH];
    int ox, oy; // global cell index of patch (0,0)
};
PARC_HD int cell_of(float p, float mn, float d) { // terrain_util.py:146-152 nearest cell, unclamped
    float f = rintf((p - mn) * DYN_RCP(d));
    f = f < -1.0e9f ? -1.0e9f : (f > 1.0e9f ? 1.0e9f : f);
    return (int)f;
}
PARC_HD float hf_at(const DynTerrain &T, int ix, int iy) {
    ix = ix < 0 ? 0 : (ix > T.X - 1 ? T.X - 1 : ix);
    iy = iy < 0 ? 0 : (iy > T.Y - 1 ? T.Y - 1 : iy);
    return T.hf[(size_t)ix * T.Y + iy];
}
PARC_HD void load_patch(const DynTerrain &T, float gx, float gy, Patch &P) {
    P.ox = cell_of(gx, T.min_x, T.dx) - DYN_PATCH / 2;
    P.oy = cell_of(gy, T.min_y, T.dy) - DYN_PATCH / 2;
    for (int a = 0; a < DYN_PATCH; ++a)
        for (int b = 0; b < DYN_PATCH; ++b) P.h[a * DYN_PATCH + b] = hf_at(T, P.ox + a, P.oy + b);
}
PARC_HD float patch_h(const DynTerrain &T, const Patch &P, int ix, int iy) {
    int a = ix - P.ox, b = iy - P.oy;
    if (a >= 0 && a < DYN_PATCH && b >= 0 && b < DYN_PATCH) return P.h[a * DYN_PATCH + b];
    return hf_at(T, ix, iy);
}

// One contact candidate: sphere (centre s in GLOBAL coordinates, radius r) against the column of cell (ix,iy).
// Returns penetration depth (>0 = contact) and the outward normal.
PARC_HD float sphere_vs_column(const DynTerrain &T, v3 s, float r, int ix, int iy, float top, v3 &n) {
    float cx = T.min_x + (float)ix * T.dx, cy = T.min_y + (float)iy * T.dy, hx = 0.5f * T.dx, hy = 0.5f * T.dy;
    float qx = s.x < cx - hx ? cx - hx : (s.x > cx + hx ? cx + hx : s.x);
    float qy = s.y < cy - hy ? cy - hy : (s.y > cy + hy ? cy + hy : s.y);
    float qz = s.z < top ? s.z : top;
    v3 d = mk(s.x - qx, s.y - qy, s.z - qz);
    float dist2 = dot(d, d);
    if (dist2 > 1e-12f) {
        float dist = DYN_SQRT(dist2);
        n = DYN_RCP(dist) * d;
        return r - dist;
    }
    // centre inside the column: leave through the top (lateral escapes are handled by the neighbour test order)
    n = mk(0.f, 0.f, 1.f);
    return r + (top - s.z);
}

// The sphere's OWN cell (its centre lies inside the footprint of cell (ix,iy), column top `top0`).
//   centre above the surface: normal +z, pen = r - (s.z - top0);
//   centre inside the solid: it leaves along the cheapest way out -- up through the top (top0 - s.z), or sideways through a
//   face whose neighbour column is lower: distance to the face + what is left to climb on the other side.  This is what a
//   blocky mesh with real vertical wall faces does to a shape that penetrates a wall a little; pushing such a point out
//   through the top (metres away for a wall) launched bodies that had crossed a wall face or touched a step riser.
// `top_of(ox, oy)` returns the top of the neighbour column at cell offset (ox, oy).
template <class TopOf>
PARC_HD float own_column_contact(const DynTerrain &T, v3 s, float r, int ix, int iy, float top0, TopOf top_of, v3 &n) {
    n = mk(0.f, 0.f, 1.f);
    if (s.z >= top0) return r - (s.z - top0);
    float best = top0 - s.z;
    const float ex = s.x - (T.min_x + (float)ix * T.dx), ey = s.y - (T.min_y + (float)iy * T.dy);
    const float hx = 0.5f * T.dx, hy = 0.5f * T.dy;
    float pen = best;
    {   // a side exit costs at least the distance to the nearest face: farther from every face than it is deep -> up, and the
        // neighbour heights need not be looked at (the usual case: a foot corner a few millimetres into the ground)
        const float dx_ = hx - (ex < 0.f ? -ex : ex), dy_ = hy - (ey < 0.f ? -ey : ey);
        if ((dx_ < dy_ ? dx_ : dy_) >= best) return pen + r;
    }
    PARC_UNROLL
    for (int f = 0; f < 4; ++f) {
        const int ox = f == 0 ? 1 : (f == 1 ? -1 : 0), oy = f == 2 ? 1 : (f == 3 ? -1 : 0);
        const float tn = top_of(ox, oy);
        if (!(tn < top0 - 1e-3f)) continue;
        const float d = f == 0 ? hx - ex : (f == 1 ? hx + ex : (f == 2 ? hy - ey : hy + ey));
        const float climb = tn > s.z ? tn - s.z : 0.f;
        if (d + climb < best) { best = d + climb; pen = d; n = mk((float)ox, (float)oy, 0.f); }
    }
    return pen + r;
}

// Where a collision segment A-B (same frame as the terrain T) meets the top edge of a column.  Every convex edge of the blocky terrain
// lies on a grid line between two cells (the reference's mesh: one quad per cell plus vertical walls, terrain_util.py:1120-1184).  If
// the segment's footprint crosses such a line and the two columns there differ in height, the edge {line, z = higher top} exists, and
// the point of the segment closest to that edge LINE is where a contact would be: returns it in Q.  The caller then tests a sphere of
// the segment's radius centred at Q against the columns around it like any other collision point (for an axis point outside the
// solid that is the sphere-vs-edge distance, exact; for one inside -- a radius-0 sole edge cutting the corner -- the own-column rule).
// One candidate per segment: per axis the crossed line next to the midpoint's cell, of the two axes the candidate closer to its edge.
// The ends of a segment are collision points themselves (end spheres, box corners): the candidate's contact is WEIGHTED by
// w = clamp(10 min(t, 1 - t), 0, 1) of its position t along the segment -- full strength over the middle 80 %, fading to nothing at
// the ends, where the point that sits there takes over.  A hard cut-off instead makes a contact of finite strength appear when an end
// moves a millimetre past a lip (measured: the three kernels then disagree on single envs by 100 N).
// `top_at(ix, iy)` returns the column top of cell (ix, iy).  Returns w (0 = no candidate).
template <class TopAt>
PARC_HD float segment_edge_point(const DynTerrain &T, v3 A, v3 Bv, TopAt top_at, v3 &Q) {
    // Straight-line code (both axes are always evaluated, validity is a flag): on the GPU all four height look-ups of a segment are
    // then in flight together.  Horizontal coordinates in cell units: u = (p - min) / d, cell index = rint(u).
    const v3 d = Bv - A;
    const float inv[2] = {DYN_RCP(T.dx), DYN_RCP(T.dy)}, dd[2] = {T.dx, T.dy};
    const float ua[2] = {(A.x - T.min_x) * inv[0], (A.y - T.min_y) * inv[1]}, ub[2] = {(Bv.x - T.min_x) * inv[0], (Bv.y - T.min_y) * inv[1]};
    float ts_[2], d2_[2];
    bool ok_[2];
    PARC_UNROLL
    for (int axis = 0; axis < 2; ++axis) {
        const int o = 1 - axis;
        const float ca = rintf(ua[axis]), cb = rintf(ub[axis]);
        const bool cross = ca != cb;
        const float sg = cb > ca ? 1.f : -1.f;
        const float cm = rintf(0.5f * (ua[axis] + ub[axis]));
        const float c0 = cm == cb ? cb - sg : cm;                 // the line between cells c0 and c0 + sg lies between the two ends
        const float lb = c0 + 0.5f * sg;
        const float du = ub[axis] - ua[axis];
        const float tc = cross ? (lb - ua[axis]) * DYN_RCP(du) : 0.f;
        const float po = ua[o] + tc * (ub[o] - ua[o]);            // the other horizontal coordinate where the line is crossed
        const float lim = 1.0e6f;
        const float c0c = c0 < -lim ? -lim : (c0 > lim ? lim : c0), poc = po < -lim ? -lim : (po > lim ? lim : po);
        const int i0 = (int)c0c, i1 = (int)(c0c + sg), io = (int)rintf(poc);
        const float t0 = axis == 0 ? top_at(i0, io) : top_at(io, i0), t1 = axis == 0 ? top_at(i1, io) : top_at(io, i1);
        const float df = t0 - t1;
        const bool edge = (df < 0.f ? -df : df) >= 1e-3f;         // the same height on both sides: no edge at this line
        const float top = t0 > t1 ? t0 : t1;
        const float da = du * dd[axis], ea = (ua[axis] - lb) * dd[axis], ez = A.z - top;
        const float ts = -(ea * da + ez * d.z) * DYN_RCP(da * da + d.z * d.z);
        const float qa = ea + ts * da, qz = ez + ts * d.z;
        ts_[axis] = ts; d2_[axis] = qa * qa + qz * qz;
        ok_[axis] = cross && edge && ts > 0.f && ts < 1.f;
    }
    const bool use1 = ok_[1] && (!ok_[0] || d2_[1] < d2_[0]);
    const float ts = use1 ? ts_[1] : ts_[0];
    Q = A + ts * d;
    float w = 10.f * (ts < 1.f - ts ? ts : 1.f - ts);
    w = w > 1.f ? 1.f : w;
    return (ok_[0] || ok_[1]) ? w : 0.f;
}

// ---------------------------------------------------------------- contact manifold
#define DYN_MAXM 128  // planes of one env (reference statement; the wave kernel keeps per-wave lists in LDS with an overflow area, see there)
struct ContactPlane {
    int body;
    v3 p;      // the contact point of the candidate sphere's centre, BODY frame
    v3 n;      // unit normal, world axes (the terrain does not move)
    float off; // pen(g) = off - n . g for the sphere centre at g (local frame of the build: see dyn_control_step)
    float w;   // stiffness / damping weight (1 for a collision point; the position weight of a segment's edge candidate)
};
struct Manifold { int n; ContactPlane e[DYN_MAXM]; };
PARC_HD float spec_margin(const DynModel &M, float vn) {
    const float m = M.spec_m0 + M.spec_tv * (vn < 0.f ? -vn : 0.f);
    return m < M.spec_max ? m : M.spec_max;
}
// Discovery for one candidate sphere (centre g in the frame of T, radius rad, velocity vpt): the column of its own cell and the higher
// neighbours (walls / step edges); emit(pen, n) for every contact within the speculative margin.  `top_at(ix, iy)` returns a column top.
template <class TopAt, class Emit>
PARC_HD void sphere_discover(const DynModel &M, const DynTerrain &T, v3 g, float rad, v3 vpt, TopAt top_at, Emit emit) {
    const int ix = cell_of(g.x, T.min_x, T.dx), iy = cell_of(g.y, T.min_y, T.dy);
    const float top0 = top_at(ix, iy);
    const float zlo = g.z - rad - spec_margin(M, vpt.z);
    for (int nb = 0; nb < 9; ++nb) {
        const int jx = ix + (nb % 3) - 1, jy = iy + (nb / 3) - 1;
        const bool own = nb == 4;
        const float top = own ? top0 : top_at(jx, jy);
        if (!own && !(top > top0 + 1e-3f)) continue; // only higher neighbours act as walls / step edges
        if (zlo > top) continue;
        v3 n;
        const float pen = own ? own_column_contact(T, g, rad, ix, iy, top0, [&](int ox, int oy) { return top_at(ix + ox, iy + oy); }, n)
                              : sphere_vs_column(T, g, rad, jx, jy, top, n);
        if (!(pen > -spec_margin(M, dot(vpt, n)))) continue;
        emit(pen, n);
    }
}

// One contact (point at x relative to O with velocity vpt, penetration pen along the unit normal n): explicit force into
// pA and the force report, implicit term dt X^T (beta 1 + (bn - beta) n n^T) X into IA.
// `w` scales the contact's stiffness and damping (1 for a collision point; the position weight of a segment's edge candidate).
PARC_HD void contact_apply(const DynModel &M, float dt, v3 x, v3 vpt, float pen, v3 n, sym6 &IA, s6 &pA, v3 &fsum, float w = 1.f) {
    const bool capped = pen > M.pen_cap;
    if (capped) pen = M.pen_cap;
    const float vn = dot(vpt, n);
    float fn = w * (M.kn * pen - M.dn * vn);
    if (fn < 0.f) fn = 0.f;
    const v3 vt = vpt - vn * n;
    const float vtm = DYN_SQRT(dot(vt, vt));
    float beta = w * M.dtang;
    if (beta * vtm > M.mu * fn) beta = vtm > 1e-9f ? M.mu * fn * DYN_RCP(vtm) : 0.f; // secant of the Coulomb cone
    const v3 f = fn * n - beta * vt;
    const v3 no = cross(x, f);
    pA.a[0] -= no.x; pA.a[1] -= no.y; pA.a[2] -= no.z; pA.a[3] -= f.x; pA.a[4] -= f.y; pA.a[5] -= f.z;
    fsum = fsum + f;
    const float bn = fn > 0.f ? w * (M.dn + (capped ? 0.f : dt * M.kn)) : 0.f; // only while pushing; a saturated spring adds no stiffness
    add_inertia(IA, dt * beta, x, nullptr);
    symrank1(IA, dt * (bn - beta), s6mk(cross(x, n), n));
}

// ---------------------------------------------------------------- per-env step
struct DynState { // pointers to this env's rows
    float *root_pos, *root_rot, *root_vel, *root_ang_vel, *dof_pos, *dof_vel;
    float *contact_force; // [B][3]
    float *body_pos;      // [B][3] or null
};

PARC_HD float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Control modes other than pd (ig_char_env.py:488-506, 374-421).  One dof's FEED-FORWARD torque, formed once per control step:
//   vel     velocity target: PhysX's drive force is damping * (target - dof_vel) with the stiffness set to 0 (:122-123): damping * target is the
//           constant part, -damping * dof_vel stays in the (implicit) drive;
//   torque  the action clipped to its bounds (= +- the motor efforts, :361-368) is the joint torque (:378-380, 498-499);
//   pd_exp, pd_1d   explicit PD torque from the state at the start of the control step (_apply_forces runs once per _step_sim, :374-397); `diff` is
//           the rotation from the joint's pose to the target's as an exponential map (pd_exp, compute_dof_vel with dt = 1, :404-409) or the plain
//           difference (pd_1d, :419); the target is the action AS GIVEN (:500-503 store `actions`, not the clipped copy); limited to the motor efforts.
PARC_HD float ctrl_ff(int ctrl, float kp, float kd, float eff, float act_lo, float act_hi, float a, float diff, float qd0) {
    if (ctrl == PARC_CTRL_VEL) return kd * clampf(a, act_lo, act_hi);
    if (ctrl == PARC_CTRL_TORQUE) return clampf(a, act_lo, act_hi);
    return clampf(kp * diff - kd * qd0, -eff, eff);
}
// a hinge's `diff`: pd_exp goes through quaternions (the shorter way round), pd_1d subtracts
PARC_HD float ctrl_hinge_diff(int ctrl, float a, float hang) {
    float d = a - hang;
    if (ctrl == PARC_CTRL_PD_EXP) d -= 6.28318530717958648f * DYN_RINT(d * 0.159154943091895336f);
    return d;
}
// ... and the substep's drive torque / joint-space augmentation with it (the pd mode's are kp err - (kd + dt kp) qd and arm + dt kd + dt^2 kp)
PARC_HD void drive_ff(int ctrl, float kd, float arm, float eff, float dt, float ff, float qd, float &t, float &aug) {
    const float kdv = ctrl == PARC_CTRL_VEL ? kd : 0.f;
    t = clampf(ff - kdv * qd, -eff, eff);
    aug = arm + dt * kdv;
}

// Advance one env by one CONTROL step (nsub solver substeps).  `action` = PD targets before clipping
// (ig_char_env.py:488-490); env_off = origin of the env in the global terrain frame.
PARC_HD void dyn_control_step(const DynModel &M, const DynTerrain &T, const DynState &S, const float *action, const float *env_off) {
    const int B = M.B;
    const float dt = M.dt;
    // ---- state into locals -----------------------------------------------------------------------------
    v3 rp = mk(S.root_pos[0], S.root_pos[1], S.root_pos[2]);
    q4 rq; rq.x = S.root_rot[0]; rq.y = S.root_rot[1]; rq.z = S.root_rot[2]; rq.w = S.root_rot[3];
    rq = qnormalize(rq);
    v3 rv = mk(S.root_vel[0], S.root_vel[1], S.root_vel[2]);
    v3 rw = mk(S.root_ang_vel[0], S.root_ang_vel[1], S.root_ang_vel[2]);
    q4 jq[DYN_MAXB];      // joint rotation parent->child (spherical / hinge)
    float hang[DYN_MAXB]; // hinge angle
    float qd[DYN_MAXD];
    q4 tq[DYN_MAXB];      // PD target rotation
    float thang[DYN_MAXB];
    for (int d = 0; d < M.D; ++d) qd[d] = S.dof_vel[d];
    float ff[DYN_MAXD];   // feed-forward joint torques of the control modes other than pd
    for (int d = 0; d < DYN_MAXD; ++d) ff[d] = 0.f;
    for (int i = 1; i < B; ++i) {
        const int di = M.dof_idx[i];
        jq[i].x = 0.f; jq[i].y = 0.f; jq[i].z = 0.f; jq[i].w = 1.f; hang[i] = 0.f; tq[i] = jq[i]; thang[i] = 0.f;
        if (M.jtype[i] == DJ_SPHERICAL) {
            jq[i] = qexp(mk(S.dof_pos[di], S.dof_pos[di + 1], S.dof_pos[di + 2]));
            if (M.ctrl == PARC_CTRL_PD) {
                tq[i] = qexp(mk(clampf(action[di], M.act_lo[di], M.act_hi[di]), clampf(action[di + 1], M.act_lo[di + 1], M.act_hi[di + 1]),
                                clampf(action[di + 2], M.act_lo[di + 2], M.act_hi[di + 2])));
            } else {
                v3 diff = mk(0.f, 0.f, 0.f);
                if (M.ctrl >= PARC_CTRL_PD_EXP) diff = qlog(qmul(qconj(jq[i]), qexp(mk(action[di], action[di + 1], action[di + 2]))));
                const float d3[3] = {diff.x, diff.y, diff.z};
                for (int k = 0; k < 3; ++k)
                    ff[di + k] = ctrl_ff(M.ctrl, M.kp[di + k], M.kd[di + k], M.eff[di + k], M.act_lo[di + k], M.act_hi[di + k], action[di + k], d3[k], qd[di + k]);
            }
        } else if (M.jtype[i] == DJ_HINGE) {
            hang[i] = S.dof_pos[di];
            if (M.ctrl == PARC_CTRL_PD) thang[i] = clampf(action[di], M.act_lo[di], M.act_hi[di]);
            else ff[di] = ctrl_ff(M.ctrl, M.kp[di], M.kd[di], M.eff[di], M.act_lo[di], M.act_hi[di], action[di], ctrl_hinge_diff(M.ctrl, action[di], hang[i]), qd[di]);
        }
    }
    Patch patch;
    load_patch(T, rp.x + env_off[0], rp.y + env_off[1], patch);
    v3 fcon[DYN_MAXB]; // net contact force of the last substep
    Manifold man; man.n = 0;
    v3 rp_d = rp;      // root position at the last discovery

    for (int sub = 0; sub < M.nsub; ++sub) {
        // ---- kinematics relative to O = root origin --------------------------------------------------------
        q4 bq[DYN_MAXB]; m3 R[DYN_MAXB]; v3 r[DYN_MAXB]; s6 vel[DYN_MAXB]; s6 cJ[DYN_MAXB];
        bq[0] = rq; R[0] = qmat(rq); r[0] = mk(0.f, 0.f, 0.f);
        vel[0] = s6mk(rw, rv); // reference point O is the root origin: v_O = root linear velocity
        cJ[0] = s6zero();
        for (int i = 1; i < B; ++i) {
            const int p = M.parent[i], di = M.dof_idx[i];
            r[i] = r[p] + mulv(R[p], mk(M.lt[i][0], M.lt[i][1], M.lt[i][2]));
            q4 lq; lq.x = M.lr[i][0]; lq.y = M.lr[i][1]; lq.z = M.lr[i][2]; lq.w = M.lr[i][3];
            q4 jr = jq[i];
            if (M.jtype[i] == DJ_HINGE) jr = qexp(hang[i] * mk(M.axis[i][0], M.axis[i][1], M.axis[i][2]));
            bq[i] = qnormalize(qmul(bq[p], qmul(lq, jr)));
            R[i] = qmat(bq[i]);
            v3 wj = mk(0.f, 0.f, 0.f);
            if (M.jtype[i] == DJ_SPHERICAL) wj = mulv(R[i], mk(qd[di], qd[di + 1], qd[di + 2]));
            else if (M.jtype[i] == DJ_HINGE) wj = qd[di] * mulv(R[i], mk(M.axis[i][0], M.axis[i][1], M.axis[i][2]));
            s6 vJ = s6mk(wj, cross(r[i], wj));
            vel[i] = vel[p] + vJ;
            cJ[i] = crm(vel[i], vJ);
        }
        // ---- articulated inertias / bias forces ------------------------------------------------------------
        sym6 IA[DYN_MAXB]; s6 pA[DYN_MAXB];
        for (int i = 0; i < B; ++i) {
            for (int k = 0; k < 21; ++k) IA[i].s[k] = 0.f;
            v3 c = r[i] + mulv(R[i], mk(M.com[i][0], M.com[i][1], M.com[i][2]));
            // Ic world = R I R^T
            float Ib[3][3] = {{M.inertia[i][0], M.inertia[i][3], M.inertia[i][4]}, {M.inertia[i][3], M.inertia[i][1], M.inertia[i][5]},
                              {M.inertia[i][4], M.inertia[i][5], M.inertia[i][2]}};
            float RI[3][3], Iw[3][3];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) RI[a][b] = R[i].m[a][0] * Ib[0][b] + R[i].m[a][1] * Ib[1][b] + R[i].m[a][2] * Ib[2][b];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Iw[a][b] = RI[a][0] * R[i].m[b][0] + RI[a][1] * R[i].m[b][1] + RI[a][2] * R[i].m[b][2];
            const float Ic[6] = {Iw[0][0], Iw[1][1], Iw[2][2], Iw[0][1], Iw[0][2], Iw[1][2]};
            add_inertia(IA[i], M.mass[i], c, Ic);
            s6 Iv = symmul(IA[i], vel[i]);
            pA[i] = crf(vel[i], Iv);
            // gravity at the COM
            v3 fg = mk(M.mass[i] * M.ext_acc[0], M.mass[i] * M.ext_acc[1], M.mass[i] * M.gravity_z);
            v3 ng = cross(c, fg);
            pA[i].a[0] -= ng.x; pA[i].a[1] -= ng.y; pA[i].a[2] -= ng.z; pA[i].a[3] -= fg.x; pA[i].a[4] -= fg.y; pA[i].a[5] -= fg.z;
            fcon[i] = mk(0.f, 0.f, 0.f);
        }
        // ---- contacts: explicit force + implicit (dt*B) inertia term ----------------------------------------
        // Discovery (substeps 0, man_period, ...): every collision point and every segment's edge candidate against the columns it can
        // reach within the speculative margin -> planes.  The planes live in the ROOT-AT-DISCOVERY frame (world axes, origin = the root
        // position rp_d of that substep): the coordinates n . g stay small (the env-local frame reaches ~1 km at 65 536 envs).
        if (sub % M.man_period == 0) {
            man.n = 0; rp_d = rp;
            const v3 goff = mk(rp.x + env_off[0], rp.y + env_off[1], rp.z + env_off[2]);
            auto top_at = [&](int ix, int iy) { return patch_h(T, patch, ix, iy); };
            auto candidate = [&](int i, v3 x, float rad, float w) {
                const v3 vpt = s6lin(vel[i]) + cross(s6ang(vel[i]), x);
                sphere_discover(M, T, x + goff, rad, vpt, top_at, [&](float pen, v3 n) {
                    if (man.n >= DYN_MAXM) return;
                    ContactPlane &P = man.e[man.n++];
                    P.body = i; P.p = mulTv(R[i], x - r[i]); P.n = n; P.off = pen + dot(n, x); P.w = w;
                });
            };
            for (int k = 0; k < M.ncol; ++k) {
                const int i = M.col_body[k];
                candidate(i, r[i] + mulv(R[i], mk(M.col_pos[k][0], M.col_pos[k][1], M.col_pos[k][2])), M.col_r[k], 1.f);
            }
            // shafts of capsules / sole edges of boxes against the top edges of the columns (segment_edge_point)
            for (int k = 0; k < M.nseg; ++k) {
                const int i = M.seg_body[k];
                const v3 xa = r[i] + mulv(R[i], mk(M.seg_a[k][0], M.seg_a[k][1], M.seg_a[k][2]));
                const v3 xb = r[i] + mulv(R[i], mk(M.seg_b[k][0], M.seg_b[k][1], M.seg_b[k][2]));
                v3 Q;
                const float w = segment_edge_point(T, xa + goff, xb + goff, top_at, Q);
                if (w > 0.f) candidate(i, Q - goff, M.seg_r[k], w);
            }
        }
        // Evaluation (every substep): penetration and velocity along the cached planes
        {
            const v3 shift = rp - rp_d;
            for (int c = 0; c < man.n; ++c) {
                const ContactPlane &P = man.e[c];
                const int i = P.body;
                const v3 x = r[i] + mulv(R[i], P.p);
                const float pen = P.off - dot(P.n, x + shift);
                if (!(pen > 0.f)) continue;
                const v3 vpt = s6lin(vel[i]) + cross(s6ang(vel[i]), x);
                contact_apply(M, dt, x, vpt, pen, P.n, IA[i], pA[i], fcon[i], P.w);
            }
        }
        // ---- inward pass ------------------------------------------------------------------------------------
        float Dinv[DYN_MAXB][6]; // symmetric 3x3 inverse: xx yy zz xy xz yz (hinge uses [0])
        s6 U[DYN_MAXB][3];
        float uu[DYN_MAXB][3];
        s6 Scol[DYN_MAXB][3];
        for (int i = B - 1; i >= 1; --i) {
            const int p = M.parent[i], di = M.dof_idx[i];
            const int nd = M.jtype[i] == DJ_SPHERICAL ? 3 : (M.jtype[i] == DJ_HINGE ? 1 : 0);
            if (nd == 0) {
                s6 Ic = symmul(IA[i], cJ[i]);
                for (int k = 0; k < 21; ++k) IA[p].s[k] += IA[i].s[k];
                pA[p] = pA[p] + pA[i] + Ic;
                continue;
            }
            // motion subspace columns, drive torques
            float tau[3] = {0.f, 0.f, 0.f}, aug[3] = {0.f, 0.f, 0.f};
            if (nd == 3) {
                for (int k = 0; k < 3; ++k) {
                    v3 a = mk(R[i].m[0][k], R[i].m[1][k], R[i].m[2][k]);
                    Scol[i][k] = s6mk(a, cross(r[i], a));
                }
                v3 err = qlog(qmul(qconj(jq[i]), tq[i])); // child frame, like _calc_pd_exp_torque ig_char_env.py:536-548
                v3 cur = qlog(jq[i]);
                const float e3[3] = {err.x, err.y, err.z}, c3[3] = {cur.x, cur.y, cur.z};
                for (int k = 0; k < 3; ++k) {
                    float t = M.kp[di + k] * e3[k] - (M.kd[di + k] + dt * M.kp[di + k]) * qd[di + k];
                    t = clampf(t, -M.eff[di + k], M.eff[di + k]);
                    aug[k] = M.arm[di + k] + dt * M.kd[di + k] + dt * dt * M.kp[di + k];
                    if (M.ctrl != PARC_CTRL_PD) drive_ff(M.ctrl, M.kd[di + k], M.arm[di + k], M.eff[di + k], dt, ff[di + k], qd[di + k], t, aug[k]);
                    if (c3[k] < M.lo[di + k]) { t += M.lim_k * (M.lo[di + k] - c3[k]) - M.lim_d * qd[di + k]; aug[k] += dt * M.lim_d + dt * dt * M.lim_k; }
                    else if (c3[k] > M.hi[di + k]) { t += M.lim_k * (M.hi[di + k] - c3[k]) - M.lim_d * qd[di + k]; aug[k] += dt * M.lim_d + dt * dt * M.lim_k; }
                    tau[k] = t;
                }
            } else {
                v3 a = mulv(R[i], mk(M.axis[i][0], M.axis[i][1], M.axis[i][2]));
                Scol[i][0] = s6mk(a, cross(r[i], a));
                float t = M.kp[di] * (thang[i] - hang[i]) - (M.kd[di] + dt * M.kp[di]) * qd[di];
                t = clampf(t, -M.eff[di], M.eff[di]);
                aug[0] = M.arm[di] + dt * M.kd[di] + dt * dt * M.kp[di];
                if (M.ctrl != PARC_CTRL_PD) drive_ff(M.ctrl, M.kd[di], M.arm[di], M.eff[di], dt, ff[di], qd[di], t, aug[0]);
                if (hang[i] < M.lo[di]) { t += M.lim_k * (M.lo[di] - hang[i]) - M.lim_d * qd[di]; aug[0] += dt * M.lim_d + dt * dt * M.lim_k; }
                else if (hang[i] > M.hi[di]) { t += M.lim_k * (M.hi[di] - hang[i]) - M.lim_d * qd[di]; aug[0] += dt * M.lim_d + dt * dt * M.lim_k; }
                tau[0] = t;
            }
            float Dm[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
            for (int k = 0; k < nd; ++k) {
                U[i][k] = symmul(IA[i], Scol[i][k]);
                float sp = 0.f;
                for (int a = 0; a < 6; ++a) sp += Scol[i][k].a[a] * pA[i].a[a];
                uu[i][k] = tau[k] - sp;
            }
            for (int k = 0; k < nd; ++k)
                for (int l = 0; l < nd; ++l) {
                    float acc = 0.f;
                    for (int a = 0; a < 6; ++a) acc += Scol[i][k].a[a] * U[i][l].a[a];
                    Dm[k][l] = acc + (k == l ? aug[k] : 0.f);
                }
            if (nd == 1) {
                Dinv[i][0] = 1.f / Dm[0][0];
            } else { // symmetric 3x3 inverse by cofactors
                float c00 = Dm[1][1] * Dm[2][2] - Dm[1][2] * Dm[2][1], c01 = Dm[0][2] * Dm[2][1] - Dm[0][1] * Dm[2][2],
                      c02 = Dm[0][1] * Dm[1][2] - Dm[0][2] * Dm[1][1];
                float det = Dm[0][0] * c00 + Dm[1][0] * c01 + Dm[2][0] * c02;
                float id = 1.f / det;
                Dinv[i][0] = c00 * id; Dinv[i][3] = c01 * id; Dinv[i][4] = c02 * id;
                Dinv[i][1] = (Dm[0][0] * Dm[2][2] - Dm[0][2] * Dm[2][0]) * id;
                Dinv[i][5] = (Dm[0][2] * Dm[1][0] - Dm[0][0] * Dm[1][2]) * id;
                Dinv[i][2] = (Dm[0][0] * Dm[1][1] - Dm[0][1] * Dm[1][0]) * id;
            }
            // K = U Dinv ; Ia = IA - K U^T ; pa = pA + Ia c + K u
            s6 Kc[3];
            float Ku[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (nd == 1) {
                for (int a = 0; a < 6; ++a) { Kc[0].a[a] = U[i][0].a[a] * Dinv[i][0]; Ku[a] = Kc[0].a[a] * uu[i][0]; }
            } else {
                const float Di[3][3] = {{Dinv[i][0], Dinv[i][3], Dinv[i][4]}, {Dinv[i][3], Dinv[i][1], Dinv[i][5]}, {Dinv[i][4], Dinv[i][5], Dinv[i][2]}};
                for (int k = 0; k < 3; ++k)
                    for (int a = 0; a < 6; ++a) Kc[k].a[a] = U[i][0].a[a] * Di[0][k] + U[i][1].a[a] * Di[1][k] + U[i][2].a[a] * Di[2][k];
                for (int a = 0; a < 6; ++a) Ku[a] = Kc[0].a[a] * uu[i][0] + Kc[1].a[a] * uu[i][1] + Kc[2].a[a] * uu[i][2];
            }
            sym6 Ia = IA[i];
            for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b) {
                    float acc = 0.f;
                    for (int k = 0; k < nd; ++k) acc += Kc[k].a[a] * U[i][k].a[b];
                    Ia.s[sidx(a, b)] -= acc;
                }
            s6 Iac = symmul(Ia, cJ[i]);
            for (int k = 0; k < 21; ++k) IA[p].s[k] += Ia.s[k];
            for (int a = 0; a < 6; ++a) pA[p].a[a] += pA[i].a[a] + Iac.a[a] + Ku[a];
        }
        // ---- root: solve IA0 a0 = -pA0 (Cholesky, SPD) --------------------------------------------------------
        s6 acc[DYN_MAXB];
        {
            float L[6][6];
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) L[a][b] = 0.f;
            for (int j = 0; j < 6; ++j) {
                float sd = sget(IA[0], j, j);
                for (int k = 0; k < j; ++k) sd -= L[j][k] * L[j][k];
                sd = sd > 1e-12f ? sqrtf(sd) : 1e3f; // a non-positive pivot is a numerical breakdown: treat the direction as immovable (no acceleration) rather than as massless
                L[j][j] = sd;
                for (int a = j + 1; a < 6; ++a) {
                    float sa = sget(IA[0], a, j);
                    for (int k = 0; k < j; ++k) sa -= L[a][k] * L[j][k];
                    L[a][j] = sa / sd;
                }
            }
            float y[6], xs[6];
            for (int a = 0; a < 6; ++a) { float sa = -pA[0].a[a]; for (int k = 0; k < a; ++k) sa -= L[a][k] * y[k]; y[a] = sa / L[a][a]; }
            for (int a = 5; a >= 0; --a) { float sa = y[a]; for (int k = a + 1; k < 6; ++k) sa -= L[k][a] * xs[k]; xs[a] = sa / L[a][a]; }
            for (int a = 0; a < 6; ++a) acc[0].a[a] = xs[a];
        }
        // ---- outward pass: joint accelerations ------------------------------------------------------------------
        float qdd[DYN_MAXD];
        for (int i = 1; i < B; ++i) {
            const int p = M.parent[i], di = M.dof_idx[i];
            const int nd = M.jtype[i] == DJ_SPHERICAL ? 3 : (M.jtype[i] == DJ_HINGE ? 1 : 0);
            s6 ap = acc[p] + cJ[i];
            if (nd == 0) { acc[i] = ap; continue; }
            float rhs[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < nd; ++k) {
                float ua = 0.f;
                for (int a = 0; a < 6; ++a) ua += U[i][k].a[a] * ap.a[a];
                rhs[k] = uu[i][k] - ua;
            }
            float q3[3] = {0.f, 0.f, 0.f};
            if (nd == 1) q3[0] = Dinv[i][0] * rhs[0];
            else {
                q3[0] = Dinv[i][0] * rhs[0] + Dinv[i][3] * rhs[1] + Dinv[i][4] * rhs[2];
                q3[1] = Dinv[i][3] * rhs[0] + Dinv[i][1] * rhs[1] + Dinv[i][5] * rhs[2];
                q3[2] = Dinv[i][4] * rhs[0] + Dinv[i][5] * rhs[1] + Dinv[i][2] * rhs[2];
            }
            acc[i] = ap;
            for (int k = 0; k < nd; ++k) {
                qdd[di + k] = q3[k];
                for (int a = 0; a < 6; ++a) acc[i].a[a] += Scol[i][k].a[a] * q3[k];
            }
        }
        // ---- implicit correction of the reported contact force: f_new = f_old - dt B a_point ---------------------
        // (cheap approximation: scale by the change of the normal velocity is skipped; the explicit force is reported)
        // ---- integrate ------------------------------------------------------------------------------------------
        v3 alpha = s6ang(acc[0]), aO = s6lin(acc[0]);
        v3 rv_new = rv + dt * (aO + cross(rw, rv)); // classical acceleration of the root origin
        v3 rw_new = rw + dt * alpha;
        rw_new = (1.f / (1.f + dt * M.ang_damping)) * rw_new;
        float wm = sqrtf(dot(rw_new, rw_new));
        if (wm > M.max_ang_vel) rw_new = (M.max_ang_vel / wm) * rw_new;
        rv = rv_new; rw = rw_new;
        rp = rp + dt * rv;
        rq = qnormalize(qmul(qexp(dt * rw), rq)); // world-frame angular velocity
        for (int i = 1; i < B; ++i) {
            const int di = M.dof_idx[i];
            if (M.jtype[i] == DJ_SPHERICAL) {
                for (int k = 0; k < 3; ++k) qd[di + k] = clampf(qd[di + k] + dt * qdd[di + k], -M.max_ang_vel, M.max_ang_vel);
                jq[i] = qnormalize(qmul(jq[i], qexp(dt * mk(qd[di], qd[di + 1], qd[di + 2])))); // child-frame velocity
            } else if (M.jtype[i] == DJ_HINGE) {
                qd[di] = clampf(qd[di] + dt * qdd[di], -M.max_ang_vel, M.max_ang_vel);
                hang[i] += dt * qd[di];
            }
        }
    }
    // ---- write back ---------------------------------------------------------------------------------------------
    S.root_pos[0] = rp.x; S.root_pos[1] = rp.y; S.root_pos[2] = rp.z;
    S.root_rot[0] = rq.x; S.root_rot[1] = rq.y; S.root_rot[2] = rq.z; S.root_rot[3] = rq.w;
    S.root_vel[0] = rv.x; S.root_vel[1] = rv.y; S.root_vel[2] = rv.z;
    S.root_ang_vel[0] = rw.x; S.root_ang_vel[1] = rw.y; S.root_ang_vel[2] = rw.z;
    for (int i = 1; i < B; ++i) {
        const int di = M.dof_idx[i];
        if (M.jtype[i] == DJ_SPHERICAL) {
            v3 e = qlog(jq[i]);
            S.dof_pos[di] = e.x; S.dof_pos[di + 1] = e.y; S.dof_pos[di + 2] = e.z;
            S.dof_vel[di] = qd[di]; S.dof_vel[di + 1] = qd[di + 1]; S.dof_vel[di + 2] = qd[di + 2];
        } else if (M.jtype[i] == DJ_HINGE) {
            S.dof_pos[di] = hang[i];
            S.dof_vel[di] = qd[di];
        }
    }
    for (int i = 0; i < B; ++i) { S.contact_force[3 * i] = fcon[i].x; S.contact_force[3 * i + 1] = fcon[i].y; S.contact_force[3 * i + 2] = fcon[i].z; }
}

// ---------------------------------------------------------------- host-side model construction
struct GeomIn { int body, type; float pos[3], pos2[3], size[3], density; };
enum { DG_BOX = 0, DG_SPHERE = 1, DG_CAPSULE = 2 };

// mass / COM / inertia of each body from its geoms, and the collision point set
inline void build_mass_and_collision(DynModel &M, const GeomIn *g, int ng) {
    const float PI = 3.14159265358979323846f;
    double m[DYN_MAXB] = {0}, mc[DYN_MAXB][3] = {{0}};
    struct Part { int body; double mass; double c[3]; double I[3][3]; };
    Part parts[64]; int np = 0;
    M.ncol = 0; M.nseg = 0; M.truncated = 0;
    auto add_seg = [&](int body, const double *a, const double *b, double r) {
        if (M.nseg >= DYN_MAXS) { ++M.truncated; return; }
        M.seg_body[M.nseg] = body; M.seg_r[M.nseg] = (float)r;
        for (int q = 0; q < 3; ++q) { M.seg_a[M.nseg][q] = (float)a[q]; M.seg_b[M.nseg][q] = (float)b[q]; }
        M.nseg++;
    };
    if (ng > 64) M.truncated += ng - 64;
    for (int k = 0; k < ng && np < 64; ++k) {
        Part P; P.body = g[k].body;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) P.I[a][b] = 0.0;
        if (g[k].type == DG_SPHERE) {
            double r = g[k].size[0];
            P.mass = g[k].density * 4.0 / 3.0 * PI * r * r * r;
            for (int a = 0; a < 3; ++a) { P.c[a] = g[k].pos[a]; P.I[a][a] = 0.4 * P.mass * r * r; }
            if (M.ncol < DYN_MAXC) { M.col_body[M.ncol] = P.body; for (int a = 0; a < 3; ++a) M.col_pos[M.ncol][a] = g[k].pos[a]; M.col_r[M.ncol] = (float)r; M.ncol++; }
            else ++M.truncated;
        } else if (g[k].type == DG_BOX) {
            double a = g[k].size[0], b = g[k].size[1], c = g[k].size[2];
            P.mass = g[k].density * 8.0 * a * b * c;
            for (int q = 0; q < 3; ++q) P.c[q] = g[k].pos[q];
            P.I[0][0] = P.mass / 3.0 * (b * b + c * c); P.I[1][1] = P.mass / 3.0 * (a * a + c * c); P.I[2][2] = P.mass / 3.0 * (a * a + b * b);
            if (M.ncol + 8 > DYN_MAXC) M.truncated += M.ncol + 8 - DYN_MAXC;
            for (int q = 0; q < 8 && M.ncol < DYN_MAXC; ++q) {
                M.col_body[M.ncol] = P.body;
                M.col_pos[M.ncol][0] = g[k].pos[0] + ((q & 1) ? (float)a : -(float)a);
                M.col_pos[M.ncol][1] = g[k].pos[1] + ((q & 2) ? (float)b : -(float)b);
                M.col_pos[M.ncol][2] = g[k].pos[2] + ((q & 4) ? (float)c : -(float)c);
                M.col_r[M.ncol] = 0.f; M.ncol++;
            }
            // the two LONG edges of the sole (the face at the body frame's lowest z; the MJCF's feet point their soles down in the rest pose),
            // as thin capsules: radius rho = 1 cm, axis inset by rho, so their surface touches the sole plane and the box's long sides.
            // A lip that runs across the foot -- a step approached head on -- crosses both; the short edges (toe / heel, 9 cm for the
            // humanoid) span less than the corners already resolve.  (A radius-0 segment would put its candidate exactly ON the grid line
            // that carries the lip: which of the two columns it is tested against would be decided by rounding.)
            const bool long_x = a >= b;
            const double rho = 0.01 < c ? 0.01 : c;
            for (int q = 0; q < 2; ++q) {
                const double sgn = q == 0 ? -1.0 : 1.0;
                const double e0[3] = {g[k].pos[0] + (long_x ? -(a - rho) : sgn * (a - rho)), g[k].pos[1] + (long_x ? sgn * (b - rho) : -(b - rho)), g[k].pos[2] - c + rho};
                const double e1[3] = {g[k].pos[0] + (long_x ? (a - rho) : sgn * (a - rho)), g[k].pos[1] + (long_x ? sgn * (b - rho) : (b - rho)), g[k].pos[2] - c + rho};
                add_seg(P.body, e0, e1, rho);
            }
        } else { // capsule
            double r = g[k].size[0];
            double d[3] = {g[k].pos2[0] - g[k].pos[0], g[k].pos2[1] - g[k].pos[1], g[k].pos2[2] - g[k].pos[2]};
            double L = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double mcyl = g[k].density * PI * r * r * L, msph = g[k].density * 4.0 / 3.0 * PI * r * r * r;
            P.mass = mcyl + msph;
            for (int q = 0; q < 3; ++q) P.c[q] = 0.5 * (g[k].pos[q] + g[k].pos2[q]);
            double Iax = mcyl * r * r / 2.0 + msph * 0.4 * r * r;
            double Itr = mcyl * (L * L / 12.0 + r * r / 4.0) + msph * (0.4 * r * r + L * L / 4.0 + 3.0 * L * r / 8.0);
            double u[3] = {0, 0, 1};
            if (L > 1e-9) for (int q = 0; q < 3; ++q) u[q] = d[q] / L;
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) P.I[a][b] = (a == b ? Itr : 0.0) + (Iax - Itr) * u[a] * u[b];
            if (M.ncol + 2 > DYN_MAXC) M.truncated += M.ncol + 2 - DYN_MAXC;
            for (int e = 0; e < 2 && M.ncol < DYN_MAXC; ++e) {
                M.col_body[M.ncol] = P.body;
                for (int q = 0; q < 3; ++q) M.col_pos[M.ncol][q] = e ? g[k].pos2[q] : g[k].pos[q];
                M.col_r[M.ncol] = (float)r; M.ncol++;
            }
            // the shaft between the two end spheres; a capsule whose shaft is about as short as its diameter (the humanoid's clavicles) is covered by them
            const double e0[3] = {g[k].pos[0], g[k].pos[1], g[k].pos[2]}, e1[3] = {g[k].pos2[0], g[k].pos2[1], g[k].pos2[2]};
            if (L > 2.0 * r + 0.02) add_seg(P.body, e0, e1, r);
        }
        m[P.body] += P.mass;
        for (int q = 0; q < 3; ++q) mc[P.body][q] += P.mass * P.c[q];
        parts[np++] = P;
    }
    M.total_mass = 0.f;
    for (int b = 0; b < M.B; ++b) {
        double mm = m[b] > 1e-9 ? m[b] : 1e-3;
        double c[3] = {mc[b][0] / mm, mc[b][1] / mm, mc[b][2] / mm};
        double I[3][3] = {{0}};
        for (int k = 0; k < np; ++k) if (parts[k].body == b) {
            double d[3] = {parts[k].c[0] - c[0], parts[k].c[1] - c[1], parts[k].c[2] - c[2]};
            double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            for (int a = 0; a < 3; ++a) for (int q = 0; q < 3; ++q) I[a][q] += parts[k].I[a][q] + parts[k].mass * ((a == q ? dd : 0.0) - d[a] * d[q]);
        }
        if (m[b] <= 1e-9) for (int a = 0; a < 3; ++a) I[a][a] = 1e-5;
        M.mass[b] = (float)mm; M.total_mass += (float)mm;
        for (int q = 0; q < 3; ++q) M.com[b][q] = (float)c[q];
        M.inertia[b][0] = (float)I[0][0]; M.inertia[b][1] = (float)I[1][1]; M.inertia[b][2] = (float)I[2][2];
        M.inertia[b][3] = (float)I[0][1]; M.inertia[b][4] = (float)I[0][2]; M.inertia[b][5] = (float)I[1][2];
    }
}

// DynModel from the C-ABI config (character tables + ParcDynamicsParams)
inline void fill_dyn_model(DynModel &M, const ParcCharModel &cm, const ParcDynamicsParams &dp, const float *act_lo, const float *act_hi) {
    M.B = cm.num_bodies; M.D = cm.dof_size;
    for (int b = 0; b < DYN_MAXB; ++b) {
        M.parent[b] = b < M.B ? cm.parent[b] : -1;
        M.jtype[b] = b < M.B ? cm.joint_type[b] : DJ_FIXED;
        M.dof_idx[b] = b < M.B ? cm.dof_idx[b] : 0;
        for (int k = 0; k < 3; ++k) { M.lt[b][k] = b < M.B ? cm.local_translation[b][k] : 0.f; M.axis[b][k] = b < M.B ? cm.joint_axis[b][k] : 0.f; }
        for (int k = 0; k < 4; ++k) M.lr[b][k] = b < M.B ? cm.local_rotation[b][k] : (k == 3 ? 1.f : 0.f);
    }
    for (int d = 0; d < DYN_MAXD; ++d) {
        const bool ok = d < M.D;
        M.kp[d] = ok ? dp.dof_stiffness[d] : 0.f; M.kd[d] = ok ? dp.dof_damping[d] : 0.f; M.arm[d] = ok ? dp.dof_armature[d] : 0.f;
        M.eff[d] = ok ? dp.dof_effort[d] : 0.f; M.lo[d] = ok ? dp.dof_lower[d] : 0.f; M.hi[d] = ok ? dp.dof_upper[d] : 0.f;
        M.act_lo[d] = ok ? act_lo[d] : 0.f; M.act_hi[d] = ok ? act_hi[d] : 0.f;
    }
    GeomIn g[PARC_MAX_GEOMS];
    const int ng = dp.num_geoms < PARC_MAX_GEOMS ? dp.num_geoms : PARC_MAX_GEOMS;
    for (int k = 0; k < ng; ++k) {
        g[k].body = dp.geom_body[k]; g[k].type = dp.geom_type[k]; g[k].density = dp.geom_density[k];
        for (int a = 0; a < 3; ++a) { g[k].pos[a] = dp.geom_pos[k][a]; g[k].pos2[a] = dp.geom_pos2[k][a]; g[k].size[a] = dp.geom_size[k][a]; }
    }
    build_mass_and_collision(M, g, ng);
    const int sub = dp.substeps > 0 ? dp.substeps : 1, steps = dp.sim_steps > 0 ? dp.sim_steps : 1;
    M.nsub = sub * steps;
    M.dt = dp.sim_dt / (float)sub;
    M.gravity_z = dp.gravity_z;
    // contact compliance (re-authored solver; PhysX's rigid contact has no such parameters)
    // Friction is a regularised Coulomb law, integrated implicitly (contact_apply: the tangential term dt * beta enters the articulated
    // inertia, so the damper itself is unconditionally stable): f_t = -beta v_t with beta = dtang while beta |v_t| <= mu f_n (STICK: the
    // implicit solve then drives the point's tangential velocity to load / beta), else beta = mu f_n / |v_t| (SLIP, the secant of the
    // cone).  dtang = 3e4 N s/m per point: a planted foot (4 sole corners, 20 kg on it) under a lateral load of half that weight creeps
    // at 0.8 mm/s (round 2, dtang = 1e4: 2.5 mm/s); the stick regime ends at |v_t| = mu f_n / dtang ~ 2 mm/s.  The value is bounded by
    // fp32: dt * dtang is a point mass of 250 kg riding on links of ~1 kg, and the joint eliminations (IA - U D^-1 U^T) cancel to what is
    // left; at dtang = 1e5 a shin + foot jammed between two walls (both contacts saturated) loses positive definiteness and the state
    // explodes within one control step (found at 16 384 envs, tests/test_dynamics_cpu.py keeps the state), 7e4 still survives that state.
    // An anchored stick spring (PhysX-style patch friction) would need per-contact state across substeps, which this kernel does not keep.
    M.kn = 5.0e4f; M.dn = 5.0e2f; M.dtang = 3.0e4f; M.mu = dp.friction;
    M.ext_acc[0] = 0.f; M.ext_acc[1] = 0.f;
    // PhysX bounds the speed at which a penetration is pushed out (max_depenetration_velocity, ig_env.py:150-160 /
    // dm_env_default.yaml sim.physx).  A spring-damper contact pushes out at v = kn pen / dn once spring and damper balance,
    // so the same bound is a cap on the penetration the spring sees: pen_cap = v_max dn / kn (0.1 m for 10 m/s).
    M.pen_cap = (dp.max_depenetration_velocity > 0.f ? dp.max_depenetration_velocity : 10.f) * M.dn / M.kn;
    // manifold: discovery once per control step; margin = PhysX's contact_offset + 1.5 x what a point covers at its approach speed until the
    // next discovery, capped so that (largest sphere + margin) stays under half a cell (the near-side-only neighbour logic of the wave kernel)
    M.ctrl = dp.control_mode;
    M.man_period = M.nsub;
    M.spec_m0 = dp.contact_offset > 0.f ? dp.contact_offset : 0.02f;
    M.spec_tv = 1.5f * (float)(M.man_period - 1) * M.dt;
    M.spec_max = 0.08f;
    M.lim_k = 1.0e3f; M.lim_d = 5.0e1f;
    M.max_ang_vel = dp.max_angular_velocity > 0.f ? dp.max_angular_velocity : 100.f;
    M.ang_damping = dp.angular_damping;
}

} // namespace parcdyn

#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
