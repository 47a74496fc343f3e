// dyn_oracle.cpp — CPU build (g++) of the re-authored dynamics core, for tests only.
// TEST INFRASTRUCTURE: compiled from the product's own portable header parc_amd/csrc/parc_dynamics.hpp so that
// (a) the invariants tests (momentum, energy, resting contact, PD response) run without a GPU and
// (b) the HIP kernel can be checked against the identical arithmetic on the host.  PhysX parity is unpinned.
#include <cstring>
#include <vector>

#include "../parc_amd/csrc/parc_dynamics.hpp"

using namespace parcdyn;

extern "C" {

void *orc_dyn_create(const ParcCharModel *cm, const ParcDynamicsParams *dp, const float *act_lo, const float *act_hi) {
    DynModel *M = new DynModel();
    memset(M, 0, sizeof(DynModel));
    fill_dyn_model(*M, *cm, *dp, act_lo, act_hi);
    return M;
}
void orc_dyn_destroy(void *m) { delete (DynModel *)m; }

void orc_dyn_set_contact(void *m, float kn, float dn, float dtang, float mu) {
    DynModel *M = (DynModel *)m;
    const float vmax = M->pen_cap * M->kn / M->dn; // keep the depenetration speed the model was built with
    M->kn = kn; M->dn = dn; M->dtang = dtang; M->mu = mu;
    M->pen_cap = vmax * dn / kn;
}
void orc_dyn_set_gravity(void *m, float g) { ((DynModel *)m)->gravity_z = g; }
void orc_dyn_set_ext_acc(void *m, float ax, float ay) { ((DynModel *)m)->ext_acc[0] = ax; ((DynModel *)m)->ext_acc[1] = ay; }
void orc_dyn_get_contact(void *m, float *out4) { DynModel *M = (DynModel *)m; out4[0] = M->kn; out4[1] = M->dn; out4[2] = M->dtang; out4[3] = M->mu; }
void orc_dyn_set_gains_scale(void *m, float s) {
    DynModel *M = (DynModel *)m;
    for (int d = 0; d < M->D; ++d) { M->kp[d] *= s; M->kd[d] *= s; }
}
void orc_dyn_get_mass(void *m, float *mass, float *com, float *inertia, float *total, int *ncol) {
    DynModel *M = (DynModel *)m;
    for (int b = 0; b < M->B; ++b) {
        mass[b] = M->mass[b];
        for (int k = 0; k < 3; ++k) com[3 * b + k] = M->com[b][k];
        for (int k = 0; k < 6; ++k) inertia[6 * b + k] = M->inertia[b][k];
    }
    *total = M->total_mass; *ncol = M->ncol;
}

// geometry of the own-cell contact on a 0.4 m grid whose cell (0,0) is centred at the origin; tops5 = own, +x, -x, +y, -y
float orc_own_column_contact(const float *tops5, float x, float y, float z, float r, float *n_out) {
    DynTerrain T; T.hf = nullptr; T.X = 1; T.Y = 1; T.min_x = 0.f; T.min_y = 0.f; T.dx = 0.4f; T.dy = 0.4f;
    v3 n;
    const float pen = own_column_contact(T, mk(x, y, z), r, 0, 0, tops5[0], [&](int ox, int oy) {
        return ox == 1 ? tops5[1] : (ox == -1 ? tops5[2] : (oy == 1 ? tops5[3] : tops5[4])); }, n);
    n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z;
    return pen;
}

// test hook: grow every collision sphere / segment by eps (PhysX's contact_offset: shapes closer than that generate a contact) -- used to
// ask "which bodies TOUCH the terrain at this pose" without integrating anything
void orc_dyn_inflate(void *m, float eps) {
    DynModel *M = (DynModel *)m;
    for (int k = 0; k < M->ncol; ++k) M->col_r[k] += eps;
    for (int k = 0; k < M->nseg; ++k) M->seg_r[k] += eps;
}
// test hook: substeps per control step (1: the reported contact force is the one evaluated AT the given pose); returns the old value
int orc_dyn_set_nsub(void *m, int n) { DynModel *M = (DynModel *)m; const int old = M->nsub; M->nsub = n; return old; }
// collision points / segments / geoms that did not fit the model's fixed tables (parc_env_create refuses a model with any)
int orc_dyn_truncated(void *m) { return ((DynModel *)m)->truncated; }
// contact discovery every n substeps (1 = every substep: the round-3 behaviour, the cached planes then never act); returns the old value
int orc_dyn_set_man_period(void *m, int n) {
    DynModel *M = (DynModel *)m; const int old = M->man_period;
    M->man_period = n < 1 ? 1 : n; M->spec_tv = 1.5f * (float)(M->man_period - 1) * M->dt;
    return old;
}
// test hook: the speculative margin of the contact discovery, margin = min(m0 + tv * approach speed, mx)
void orc_dyn_set_spec(void *m, float m0, float tv, float mx) { DynModel *M = (DynModel *)m; M->spec_m0 = m0; M->spec_tv = tv; M->spec_max = mx; }
int orc_dyn_get_nseg(void *m) { return ((DynModel *)m)->nseg; }
// counterfactual for the tests: drop the collision segments (points only, the round-2 contact geometry); returns the old count
int orc_dyn_set_nseg(void *m, int n) { DynModel *M = (DynModel *)m; const int old = M->nseg; M->nseg = n; return old; }

// geometry of segment_edge_point on a heightfield: returns 1 and Q (x, y, z, weight) when the segment A-B has an edge candidate
int orc_segment_edge_point(const float *hf, int X, int Y, float min_x, float min_y, float dx, float dy, const float *A, const float *B, float *Q_out) {
    DynTerrain T; T.hf = hf; T.X = X; T.Y = Y; T.min_x = min_x; T.min_y = min_y; T.dx = dx; T.dy = dy;
    v3 Q = mk(0.f, 0.f, 0.f);
    const float w = segment_edge_point(T, mk(A[0], A[1], A[2]), mk(B[0], B[1], B[2]), [&](int ix, int iy) { return hf_at(T, ix, iy); }, Q);
    Q_out[0] = Q.x; Q_out[1] = Q.y; Q_out[2] = Q.z; Q_out[3] = w;
    return w > 0.f ? 1 : 0;
}

// state arrays are [n][...] row-major like the env tensors
void orc_dyn_step(void *m, const float *hf, int X, int Y, float min_x, float min_y, float dx, float dy, int n, float *root_pos,
                  float *root_rot, float *root_vel, float *root_ang_vel, float *dof_pos, float *dof_vel, float *contact_force,
                  const float *action, const float *env_off) {
    const DynModel &M = *(DynModel *)m;
    DynTerrain T; T.hf = hf; T.X = X; T.Y = Y; T.min_x = min_x; T.min_y = min_y; T.dx = dx; T.dy = dy;
    for (int e = 0; e < n; ++e) {
        DynState S;
        S.root_pos = root_pos + 3 * (size_t)e; S.root_rot = root_rot + 4 * (size_t)e; S.root_vel = root_vel + 3 * (size_t)e;
        S.root_ang_vel = root_ang_vel + 3 * (size_t)e; S.dof_pos = dof_pos + (size_t)M.D * e; S.dof_vel = dof_vel + (size_t)M.D * e;
        S.contact_force = contact_force + (size_t)3 * M.B * e; S.body_pos = nullptr;
        dyn_control_step(M, T, S, action + (size_t)M.D * e, env_off + 3 * (size_t)e);
    }
}
}
