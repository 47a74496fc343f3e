// parc_dynamics_wave.hpp — wave-per-limb mapping of the dynamics step for gfx950 (HIP only).
//
// Same equations as parc_dynamics.hpp.  Mapping: a 256-thread block carries 64 envs; LANE = env, WAVE = limb.
// Wave w owns limb chain w+1 (humanoid: right arm, left arm, right leg, left leg) of all 64 envs; wave 0 also
// owns the trunk chain (pelvis, torso, head).  Consequences:
//   * every lane of a wave works on the SAME body, so body / joint / collision tables are wave-uniform: they are
//     scalar loads, the joint type is a uniform branch, collision-point loops have uniform trip counts;
//   * all 64 lanes are busy in the limb phases (the chain-parallel kernel in parc_dynamics_coop.hpp keeps 5 of 8
//     lanes busy there and 1 of 8 in the trunk phase), and the trunk phase costs one wave for 64 envs;
//   * a chain's joint state and kinematics stay in registers (k loops are fully unrolled, bodies are constants),
//     the joint-space factors K = U D^-1 and D^-1 u park in LDS between the inward and the outward pass, laid out
//     [slot][lane] so that every LDS access is conflict-free;
//   * the waves of a block meet at NO barrier inside the substep loop: every hand-off (parent kinematics, a limb's articulated inertia,
//     a trunk body's record, the attach acceleration) is a record in LDS plus a flag (producer: record, release fence, flag; consumer:
//     bounded poll with s_sleep, acquire fence, read), so a wave waits only for what it depends on.  A wait that
//     hits its bound counts into g_wave_timeouts (parc_env_dynamics_timeouts, surfaced by HipParkourEnv.get_extra_log_info and
//     bench.py) and the block writes NaN root positions for its envs, so the damage shows up in every finiteness check.
// FP CONTRACTION: this header (like parc_dynamics.hpp / parc_dynamics_coop.hpp) opts INTO `fp contract(fast)` by pragma although the
// library is built with -ffp-contract=off (which lib.load() insists on): the dynamics has no bit-exact reference (PhysX parity is
// unpinned) and is judged at the dynamics tolerances of tests/test_dynamics_*.py; the flag protects the observation / reward path,
// whose 1e-5 parity depends on ATen-like rounding.  The prep-record quaternions the epilogue forms call the parc:: functions of
// parc_math.hpp, which is parsed under the command line's -ffp-contract=off: contraction is a property of the expression where it is
// written, inlining into this kernel does not change it -- the records are bit-identical to k_env_prep's (GPU test).
// BUILD NOTE: the library is compiled with -fno-slp-vectorize.  With SLP vectorisation on, hipcc (ROCm 7.2) produces a
// k_dynamics_wave whose 6x6 inertias are wrong on gfx950 (entries from index 2 on; found by comparing the three dynamics
// kernels and the CPU build, tools/dyn_cmp.py); the scalar build agrees with them to 1e-6.  The cause was not isolated:
// packed-fp32 operand selection on SGPR pairs was checked in isolation and behaves correctly, so the suspect is the
// handling of 64-bit register pairs at this kernel's register pressure (256 VGPR + 242 AGPR).
#pragma once
#include "parc_dynamics_coop.hpp"

#if defined(__HIPCC__)
#pragma clang fp contract(fast)
#endif

namespace parcdyn {

#define WV_MAXLEN 3   // bodies per chain
#define WV_MAXLIMB 4  // limb chains = waves per block
#define WV_MAXATT 3   // trunk bodies that carry limbs
#define WV_FAC 21     // K (18) + D^-1 u (3); a hinge uses slots 0..5 (K) and 6 (D^-1 u)

// Everything the wave kernel reads about a body, contiguous (256 B): the body index is wave-uniform but dynamic, so every
// field access is a scalar load; one record per body turns ~15 separate s_load + s_waitcnt round trips per body and pass
// (one per DynModel array) into a few wide loads from one base address.
struct WvBodyC {
    int jtype, dof_idx, npt, pt0;
    float lt[3], mass;
    float lr[4];
    float axis[3], brho;
    float com[3], pad0;
    float inertia[6], pad1[2];
    float bc[3], pad2;
    float kp[3], kd[3], arm[3], eff[3], lo[3], hi[3], act_lo[3], act_hi[3]; // the joint's dofs (hinge: [0])
    int nsg, sg0;            // collision segments of the body (capsule axes / sole edges, DynModel::seg_*)
    int own_npt, child;      // a FIXED leaf body merged into this one (the humanoid's hands into the lower arms, build_wave_tables): its collision
                             // points are this body's points [own_npt, npt), its index `child` (-1: none) receives their contact force
    float pad3[4];
};
static_assert(sizeof(WvBodyC) == 256, "WvBodyC is one 256-byte record");

struct WaveTables {
    WvBodyC c[DYN_MAXB];
    float colp[DYN_MAXC][4];               // collision point: body-frame position, radius
    float seg[DYN_MAXS][8];                // collision segment: end a, radius, end b, pad (body frame)
    int nlimb;
    int len[1 + WV_MAXLIMB];               // chain 0 = trunk, 1.. = limbs
    int body[1 + WV_MAXLIMB][WV_MAXLEN];
    int par_slot[1 + WV_MAXLIMB];          // attach slot of the trunk body a limb hangs off
    int early[1 + WV_MAXLIMB];             // limb hangs off a non-root trunk body: finished before the trunk's upper part starts
    int ep_trunk_wave[WV_MAXLEN], ep_head_wave; // epilogue: which wave (>= 1) forms the prep-record quaternion of trunk position k / the heading terms
                                                // of the root from what wave 0 hands over (-1: wave 0 does it itself)
    int helper;                            // wave (>= 1) of an early limb: idle in part B, it prepares own inertia + contacts of trunk bodies; -1 if none
    int prep[WV_MAXLEN];                   // per trunk position: 1 = own inertia + contacts prepared by another wave (rec_wave), handed over as a record
    int rec_wave[WV_MAXLEN];               // that wave (>= 1), or -1: wave 0 does the body itself.  Position 0 (the root body) is prepared AFTER the
                                           // preparing wave's own limb (it is needed last), positions >= 1 between its limb's kinematics and inward pass
    int att_slot[WV_MAXLEN];               // per trunk position: attach slot or -1
    int nchild[WV_MAXLEN];                 // per trunk position: limbs hanging off it
    int child[WV_MAXLEN][WV_MAXLIMB];      // limb chain ids (1..)
    int npt[DYN_MAXB], pt0[DYN_MAXB];
    float brad[DYN_MAXB];
    float bc[DYN_MAXB][3], brho[DYN_MAXB];  // bounding sphere of a body's collision spheres (body frame): centre = middle of the
                                            // centres' box, radius includes the sphere radii
    int man_base[WV_MAXLIMB], man_cap[WV_MAXLIMB]; // per wave: first LDS slot and LDS share of its contact-plane list (WvMan), in proportion to the
                                                   // candidates (points + segments) of the bodies it discovers: its limb + the trunk records it prepares
};

// LDS layout (floats); everything is [slot][64 lanes]
#define WV_OFF_ATTKIN 0
#define WV_OFF_UP (WV_OFF_ATTKIN + WV_MAXATT * 13 * 64)
#define WV_OFF_ATTACC (WV_OFF_UP + WV_MAXLIMB * 27 * 64)
#define WV_OFF_ROOTP (WV_OFF_ATTACC + WV_MAXATT * 6 * 64)
#define WV_OFF_PATCH (WV_OFF_ROOTP + 3 * 64)
#define WV_OFF_ROOTI (WV_OFF_PATCH + DYN_PATCH * DYN_PATCH * 64)
#define WV_OFF_FLAG (WV_OFF_ROOTI + WV_MAXLEN * 30 * 64)   // hand-off flags between the waves of a block (see WV_F_*)
// A flag holds the number of substeps for which its producer has published: a consumer of substep `sub` waits for > sub.
#define WV_F_KIN(slot) (12 + (slot)) // wave 0: kinematics of attach slot (0..2) (+ the root position, with the root body's slot)
#define WV_F_UP(lc) (lc)      // limb chain lc (1..4): its articulated inertia / bias handed to the trunk
#define WV_F_REC(k) (5 + (k)) // record (own inertia + contacts) of trunk position k (0..2), prepared by wave rec_wave[k]
#define WV_F_HAND 11           // epilogue: wave 0 has handed the trunk joints' dofs to the helper wave
#define WV_F_ACC(slot) (8 + (slot)) // wave 0: spatial acceleration of attach slot (0..2)
// The contact planes of the four waves (WvMan: man_total slots of 8 x 64 floats) take the rest of the CU's LDS.  (Round 3 kept the
// joint-space factors K = U D^-1, D^-1 u of all bodies here, 50 KB, and the running maxima of the height patch, 19 KB: the limbs' factors
// are register arrays now -- a lane reads back what it wrote --, the maxima are gone, which is what makes the room.)
#ifdef PARC_TRUNK_FAC_REGS
#define WV_OFF_MAN (WV_OFF_FLAG + 64)
#else
#define WV_OFF_TFAC (WV_OFF_FLAG + 64)                        // wave 0: joint-space factors of trunk positions 1..2
#define WV_OFF_MAN (WV_OFF_TFAC + (WV_MAXLEN - 1) * WV_FAC * 64)
#endif
#define WV_LDS_BYTES (160 * 1024)
#define WV_MAN_TOTAL ((WV_LDS_BYTES / 4 - WV_OFF_MAN) / (8 * 64))
inline int wv_lds_floats() { return WV_OFF_MAN + WV_MAN_TOTAL * 8 * 64; }

inline bool build_wave_tables(const DynModel &M, const CoopTables &C, WaveTables &W) {
    memset(&W, 0, sizeof(W));
    if (C.nchain < 1 || C.nchain > 1 + WV_MAXLIMB || C.nlevel > 2) return false;
    W.nlimb = C.nchain - 1;
    for (int c = 0; c < C.nchain; ++c) {
        if (C.len[c] > WV_MAXLEN) return false;
        W.len[c] = C.len[c];
        for (int k = 0; k < C.len[c]; ++k) W.body[c][k] = C.body[c][k];
    }
    // A leaf body on a FIXED joint is a rigid part of its parent: the articulated-body recursion through a fixed joint only adds the
    // child's inertia, bias and contact wrench to the parent's.  Merging it at table-build time (composite mass, centre of mass and
    // inertia in the parent's frame; its collision points moved into the parent's frame) removes one body from the limb's kinematics,
    // own-inertia and elimination passes -- the humanoid's hands: -1 of 3 bodies on the two arm waves, of which wave 0 is the critical
    // one.  The contact force of the child is still reported on its own (the observation's contact flags are per body).
    int merged_child[DYN_MAXB];
    for (int b = 0; b < DYN_MAXB; ++b) merged_child[b] = -1;
    for (int c = 1; c < C.nchain; ++c) {
        const int L = W.len[c];
        if (L < 2) continue;
        const int bl = W.body[c][L - 1], bp = W.body[c][L - 2];
        bool leaf = M.jtype[bl] == DJ_FIXED && M.parent[bl] == bp && C.nchild[bl] == 0 && C.nsg[bl] == 0 &&
                    (C.npt[bl] == 0 || (C.npt[bp] > 0 && C.pt0[bl] == C.pt0[bp] + C.npt[bp]));
        for (int b = 0; b < M.B; ++b) if (M.parent[b] == bl) leaf = false;
        if (!leaf) continue;
        merged_child[bp] = bl;
        W.len[c] = L - 1;
    }
    int natt = 0;
    for (int k = 0; k < WV_MAXLEN; ++k) W.att_slot[k] = -1;
    for (int c = 1; c < C.nchain; ++c) {
        int pos = -1;
        for (int k = 0; k < W.len[0]; ++k) if (W.body[0][k] == C.par_body[c]) pos = k;
        if (pos < 0) return false; // limb does not hang off the trunk
        if (W.att_slot[pos] < 0) { if (natt >= WV_MAXATT) return false; W.att_slot[pos] = natt++; }
        W.par_slot[c] = W.att_slot[pos];
        W.early[c] = pos > 0;
        W.child[pos][W.nchild[pos]++] = c;
    }
    // collision tables in the frame of the body that carries them (a merged child's points move into its parent's frame)
    for (int k = 0; k < M.ncol; ++k) { for (int a = 0; a < 3; ++a) W.colp[k][a] = M.col_pos[k][a]; W.colp[k][3] = M.col_r[k]; }
    for (int k = 0; k < M.nseg; ++k) { for (int a = 0; a < 3; ++a) { W.seg[k][a] = M.seg_a[k][a]; W.seg[k][4 + a] = M.seg_b[k][a]; } W.seg[k][3] = M.seg_r[k]; }
    double mrot[DYN_MAXB][3][3];
    for (int b = 0; b < M.B; ++b) { // rotation of body b's fixed local rotation lr (xyzw), used for merged children
        const double x = M.lr[b][0], y = M.lr[b][1], z = M.lr[b][2], w_ = M.lr[b][3];
        const double Rq[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)}, {2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)},
                                 {2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)}};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mrot[b][i][j] = Rq[i][j];
    }
    for (int b = 0; b < M.B; ++b) {
        W.npt[b] = C.npt[b]; W.pt0[b] = C.pt0[b]; W.brad[b] = C.brad[b];
        const int ch = merged_child[b];
        if (ch >= 0) {
            for (int i = 0; i < C.npt[ch]; ++i) {
                const int k = C.pt0[ch] + i;
                for (int a = 0; a < 3; ++a)
                    W.colp[k][a] = (float)(M.lt[ch][a] + mrot[ch][a][0] * M.col_pos[k][0] + mrot[ch][a][1] * M.col_pos[k][1] + mrot[ch][a][2] * M.col_pos[k][2]);
            }
            if (C.npt[b] == 0) W.pt0[b] = C.pt0[ch];
            W.npt[b] = C.npt[b] + C.npt[ch];
        }
        float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < W.npt[b]; ++i)
            for (int a = 0; a < 3; ++a) {
                const float v = W.colp[W.pt0[b] + i][a];
                if (i == 0 || v < lo[a]) lo[a] = v;
                if (i == 0 || v > hi[a]) hi[a] = v;
            }
        float rho = 0.f;
        for (int a = 0; a < 3; ++a) W.bc[b][a] = 0.5f * (lo[a] + hi[a]);
        for (int i = 0; i < W.npt[b]; ++i) {
            const float *q = W.colp[W.pt0[b] + i];
            const float d = sqrtf((q[0] - W.bc[b][0]) * (q[0] - W.bc[b][0]) + (q[1] - W.bc[b][1]) * (q[1] - W.bc[b][1]) + (q[2] - W.bc[b][2]) * (q[2] - W.bc[b][2]));
            if (d + q[3] > rho) rho = d + q[3];
        }
        W.brho[b] = rho * 1.0001f + 1e-6f;
    }
    W.helper = -1;
    for (int c = 2; c < C.nchain; ++c) if (W.early[c] && W.att_slot[0] >= 0) { W.helper = c - 1; break; }
    // Who prepares the trunk bodies' records.  Wave 0 is the critical path of a substep (its limb, then the serial chain head -> torso
    // -> pelvis -> root solve: tools/wave_timeline.py), so every record it does not have to form itself shortens the step:
    //   * positions that carry limbs (pelvis, torso): the helper wave, i.e. the wave of an early limb (an arm: short chain, few contacts);
    //   * positions without limbs (the head): the wave of the LAST limb that hangs off the root (a leg: its parent kinematics arrive first),
    //     which needs an attach slot for the position's kinematics.
    for (int k = 0; k < WV_MAXLEN; ++k) W.rec_wave[k] = -1;
    for (int k = 0; k < 2 && k < W.len[0]; ++k) if (W.helper >= 0 && W.att_slot[k] >= 0) W.rec_wave[k] = W.helper;
    int root_limb = -1;
    for (int c = 2; c < C.nchain; ++c) if (!W.early[c] && c - 1 != W.helper) root_limb = c - 1;
    for (int k = 1; k < W.len[0]; ++k)
        if (W.rec_wave[k] < 0 && W.att_slot[k] < 0 && root_limb >= 1 && natt < WV_MAXATT) { W.att_slot[k] = natt++; W.rec_wave[k] = root_limb; }
    for (int k = 0; k < W.len[0]; ++k) W.prep[k] = W.rec_wave[k] >= 1 ? 1 : 0;
    // epilogue shares: wave 0 is the last one out of the substep loop, so the transcendental-heavy prep-record work on ITS state (heading
    // terms of the root, dof -> quat of the trunk joints) goes to two other waves, which are through their own stores by then
    // (one joint per wave: a spherical joint's dof -> quat is ~700 instructions that only THIS wave executes -- cold in the instruction
    // cache, 9 k cycles --; two of them on one wave made that wave the last one out, measured)
    W.ep_head_wave = W.helper;
    {
        int cand[WV_MAXLIMB], nc = 0;
        for (int c = 2; c < C.nchain; ++c) if (c - 1 != W.helper) cand[nc++] = c - 1;   // the waves that carry neither wave 0's nor the helper's share
        for (int k = 0; k < WV_MAXLEN; ++k) W.ep_trunk_wave[k] = W.helper;
        for (int k = 1, at = 0; k < W.len[0] && nc > 0 && W.helper >= 1; ++k, ++at) W.ep_trunk_wave[k] = cand[at % nc];
    }
    {   // LDS shares of the plane lists: two slots each, the rest in proportion to the candidates a wave discovers
        int cand[WV_MAXLIMB] = {0, 0, 0, 0}, tot = 0, used = 0;
        for (int w = 0; w < WV_MAXLIMB; ++w) {
            if (w < W.nlimb) for (int k = 0; k < W.len[w + 1]; ++k) cand[w] += W.npt[W.body[w + 1][k]] + C.nsg[W.body[w + 1][k]];
            for (int k = 0; k < W.len[0]; ++k) if (W.rec_wave[k] == w || (w == 0 && W.rec_wave[k] < 0)) cand[w] += W.npt[W.body[0][k]] + C.nsg[W.body[0][k]];
            tot += cand[w];
        }
        for (int w = 0; w < WV_MAXLIMB; ++w) { W.man_cap[w] = 2 + (tot > 0 ? (WV_MAN_TOTAL - 2 * WV_MAXLIMB) * cand[w] / tot : 0); used += W.man_cap[w]; }
        for (int w = 0; used < WV_MAN_TOTAL; w = (w + 1) % WV_MAXLIMB) { ++W.man_cap[w]; ++used; }
        for (int w = 0, at = 0; w < WV_MAXLIMB; ++w) { W.man_base[w] = at; at += W.man_cap[w]; }
    }
    for (int b = 0; b < M.B; ++b) {
        WvBodyC &c = W.c[b];
        c.jtype = M.jtype[b]; c.dof_idx = M.dof_idx[b]; c.npt = W.npt[b]; c.pt0 = W.pt0[b];
        c.own_npt = C.npt[b]; c.child = merged_child[b];
        c.mass = M.mass[b]; c.brho = W.brho[b];
        for (int a = 0; a < 3; ++a) { c.lt[a] = M.lt[b][a]; c.axis[a] = M.axis[b][a]; c.com[a] = M.com[b][a]; c.bc[a] = W.bc[b][a]; }
        for (int a = 0; a < 4; ++a) c.lr[a] = M.lr[b][a];
        for (int a = 0; a < 6; ++a) c.inertia[a] = M.inertia[b][a];
        if (merged_child[b] >= 0) { // composite of b and its fixed leaf: mass, centre of mass, inertia about the composite centre, b's frame
            const int ch = merged_child[b];
            const double mp = M.mass[b], mc = M.mass[ch], mt = mp + mc;
            double cc[3], cm[3];
            for (int a = 0; a < 3; ++a) cc[a] = M.lt[ch][a] + mrot[ch][a][0] * M.com[ch][0] + mrot[ch][a][1] * M.com[ch][1] + mrot[ch][a][2] * M.com[ch][2];
            for (int a = 0; a < 3; ++a) cm[a] = (mp * M.com[b][a] + mc * cc[a]) / mt;
            auto full = [](const float *i6, double I[3][3]) { I[0][0] = i6[0]; I[1][1] = i6[1]; I[2][2] = i6[2]; I[0][1] = I[1][0] = i6[3]; I[0][2] = I[2][0] = i6[4]; I[1][2] = I[2][1] = i6[5]; };
            double Ip[3][3], Ic0[3][3], Icr[3][3], tmp[3][3], It[3][3];
            full(M.inertia[b], Ip); full(M.inertia[ch], Ic0);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { tmp[i][j] = 0; for (int k = 0; k < 3; ++k) tmp[i][j] += mrot[ch][i][k] * Ic0[k][j]; }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Icr[i][j] = 0; for (int k = 0; k < 3; ++k) Icr[i][j] += tmp[i][k] * mrot[ch][j][k]; }
            const double dp[3] = {M.com[b][0] - cm[0], M.com[b][1] - cm[1], M.com[b][2] - cm[2]}, dc[3] = {cc[0] - cm[0], cc[1] - cm[1], cc[2] - cm[2]};
            const double dp2 = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2], dc2 = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
                It[i][j] = Ip[i][j] + mp * ((i == j ? dp2 : 0.0) - dp[i] * dp[j]) + Icr[i][j] + mc * ((i == j ? dc2 : 0.0) - dc[i] * dc[j]);
            c.mass = (float)mt;
            for (int a = 0; a < 3; ++a) c.com[a] = (float)cm[a];
            c.inertia[0] = (float)It[0][0]; c.inertia[1] = (float)It[1][1]; c.inertia[2] = (float)It[2][2];
            c.inertia[3] = (float)It[0][1]; c.inertia[4] = (float)It[0][2]; c.inertia[5] = (float)It[1][2];
        }
        const int nd = b == 0 ? 0 : (M.jtype[b] == DJ_SPHERICAL ? 3 : (M.jtype[b] == DJ_HINGE ? 1 : 0));
        for (int q = 0; q < nd; ++q) {
            const int d = M.dof_idx[b] + q;
            c.kp[q] = M.kp[d]; c.kd[q] = M.kd[d]; c.arm[q] = M.arm[d]; c.eff[q] = M.eff[d]; c.lo[q] = M.lo[d]; c.hi[q] = M.hi[d];
            c.act_lo[q] = M.act_lo[d]; c.act_hi[q] = M.act_hi[d];
        }
    }
    for (int b = 0; b < M.B; ++b) {
        W.c[b].nsg = C.nsg[b]; W.c[b].sg0 = C.sg0[b];
        if (W.npt[b] + C.nsg[b] > 32) return false;
    }
    return true;
}

#if defined(__HIPCC__)

__device__ unsigned int g_wave_timeouts; // flag waits that hit their bound (exact count): must stay 0, see parc_env_dynamics_timeouts
#ifndef PARC_FLAG_SPIN_BOUND
#define PARC_FLAG_SPIN_BOUND (1 << 22)  // polls of one wait before it gives up (~0.3 s)
#endif
// -DPARC_TEST_BREAK_FLAG=<flag index>: test build (parc_amd/libparc_env_breakflag.so, tests/test_dynamics_gpu.py) whose producer
// never publishes that flag -- every consumer of it runs into the bound; the shipped library has no such switch.

// Diagnostic builds only (-DPARC_STAMPS): cycles between consecutive stamp points, summed per wave role over all blocks.
// Even slots = work segments, odd slots = the barrier wait that follows (see the WSTAMP calls in the substep loop).
// -DPARC_TIMELINE: absolute cycle counter at the hand-off points of ONE substep (the third) of block 0, per wave: who waits for whom
// (tools/wave_timeline.py).  A dozen s_memtime per wave and substep; separate from PARC_STAMPS, whose per-body stamps perturb more.
#ifdef PARC_TIMELINE
__device__ unsigned long long g_wave_tl[4][16];
#define WTL(i) do { if (blockIdx.x == 0 && sub == 2) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) g_wave_tl[w][i] = t_; } } while (0)
#else
#define WTL(i) do { } while (0)
#endif

#ifdef PARC_STAMPS
__device__ unsigned long long g_wave_stamps[4][16];
__device__ unsigned long long g_wave_cnt[16][8]; // filled by -DPARC_COUNTS builds only, per body: lanes, near lanes (body window reaches down to its 5x5 maximum), waves with a
                                                 // near lane, (lane, candidate) pairs that enter the narrow phase, candidates of the body (points + segments) x waves,
                                                 // slow lanes, narrow-phase executions (a candidate with at least one lane), waves
__device__ unsigned long long g_wave_hist[16][4][16]; // -DPARC_COUNTS: substep 0, per body (15 = the wave's whole limb): histogram of the wave-maximum of the per-lane
                                                      // count of [0] flat contacts (point candidate, normal +z), [1] generic contacts, [2] both, within the speculative margin;
                                                      // [3][0..2] = sums over lanes of flat / generic / lanes
#define WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); wacc[i] += t_ - wlast; wlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
// pin: the 27 values of (IA, pA) must be complete before the stamp that follows (keeps arithmetic from sinking past it)
#define WPIN(IA_, pA_) do { for (int i_ = 0; i_ < 21; ++i_) asm volatile("" : "+v"((IA_).s[i_])); for (int i_ = 0; i_ < 6; ++i_) asm volatile("" : "+v"((pA_).a[i_])); } while (0)
#define WSTAMP_PARAMS , unsigned long long *wacc, unsigned long long &wlast
#define WSTAMP_ARGS , wacc, wlast
#else
#define WSTAMP(i) do { } while (0)
#define WPIN(IA_, pA_) do { } while (0)
#define WSTAMP_PARAMS
#define WSTAMP_ARGS
#endif

struct WvBody { // per-body registers of the owning lane
    q4 jq, tq; float hang, thang; v3 qd; // joint state (spherical: quaternion + child-frame omega; hinge: angle + rate)
    q4 bq; v3 r; s6 vel, cJ;             // kinematics of the current substep (common frame, origin = root)
    v3 fcon, qdd;
    v3 fcon_c;                           // contact force on the fixed leaf merged into this body (WvBodyC::child)
};

template <bool FF> // FF: a control mode other than pd -- B.tq.xyz / B.thang then hold the joint's feed-forward torque (ctrl_ff) instead of the target
__device__ __forceinline__ void wv_load_joint(const DynModel &M, const WaveTables &W, int b, WvBody &B, const float *dp, const float *dv, const float *ac) {
    B.jq.x = 0.f; B.jq.y = 0.f; B.jq.z = 0.f; B.jq.w = 1.f; B.tq = B.jq; B.hang = 0.f; B.thang = 0.f; B.qd = mk(0.f, 0.f, 0.f);
    B.fcon = mk(0.f, 0.f, 0.f); B.qdd = mk(0.f, 0.f, 0.f); B.fcon_c = mk(0.f, 0.f, 0.f);
    const int jt = W.c[b].jtype, di = W.c[b].dof_idx;
    if (jt == DJ_SPHERICAL) {
        B.jq = qexp(mk(dp[di], dp[di + 1], dp[di + 2]));
        B.qd = mk(dv[di], dv[di + 1], dv[di + 2]);
        if (!FF) {
            B.tq = qexp(mk(clampf(ac[di], W.c[b].act_lo[0], W.c[b].act_hi[0]), clampf(ac[di + 1], W.c[b].act_lo[1], W.c[b].act_hi[1]),
                           clampf(ac[di + 2], W.c[b].act_lo[2], W.c[b].act_hi[2])));
        } else {
            v3 diff = mk(0.f, 0.f, 0.f);
            if (M.ctrl >= PARC_CTRL_PD_EXP) diff = qlog(qmul(qconj(B.jq), qexp(mk(ac[di], ac[di + 1], ac[di + 2]))));
            B.tq.x = ctrl_ff(M.ctrl, W.c[b].kp[0], W.c[b].kd[0], W.c[b].eff[0], W.c[b].act_lo[0], W.c[b].act_hi[0], ac[di], diff.x, B.qd.x);
            B.tq.y = ctrl_ff(M.ctrl, W.c[b].kp[1], W.c[b].kd[1], W.c[b].eff[1], W.c[b].act_lo[1], W.c[b].act_hi[1], ac[di + 1], diff.y, B.qd.y);
            B.tq.z = ctrl_ff(M.ctrl, W.c[b].kp[2], W.c[b].kd[2], W.c[b].eff[2], W.c[b].act_lo[2], W.c[b].act_hi[2], ac[di + 2], diff.z, B.qd.z);
        }
    } else if (jt == DJ_HINGE) {
        B.hang = dp[di]; B.qd.x = dv[di];
        if (!FF) B.thang = clampf(ac[di], W.c[b].act_lo[0], W.c[b].act_hi[0]);
        else B.thang = ctrl_ff(M.ctrl, W.c[b].kp[0], W.c[b].kd[0], W.c[b].eff[0], W.c[b].act_lo[0], W.c[b].act_hi[0], ac[di], ctrl_hinge_diff(M.ctrl, ac[di], B.hang), B.qd.x);
    }
}

// kinematics of body b given its parent's (pq, pr, pv); leaves its own in (pq, pr, pv) for the next body of the chain
__device__ __forceinline__ void wv_fk_body(const DynModel &M, const WaveTables &W, int b, WvBody &B, q4 &pq, v3 &pr, s6 &pv) {
    s6 cJ = s6zero();
    if (b != 0) {
        const int jt = W.c[b].jtype;
        const m3 Rp = qmat(pq);
        pr = pr + mulv(Rp, mk(W.c[b].lt[0], W.c[b].lt[1], W.c[b].lt[2]));
        q4 lq; lq.x = W.c[b].lr[0]; lq.y = W.c[b].lr[1]; lq.z = W.c[b].lr[2]; lq.w = W.c[b].lr[3];
        q4 jr = B.jq;
        if (jt == DJ_HINGE) jr = qexp(B.hang * mk(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]));
        pq = qnormalize(qmul(pq, qmul(lq, jr)));
        const m3 R = qmat(pq);
        v3 wj = mk(0.f, 0.f, 0.f);
        if (jt == DJ_SPHERICAL) wj = mulv(R, B.qd);
        else if (jt == DJ_HINGE) wj = B.qd.x * mulv(R, mk(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]));
        const s6 vJ = s6mk(wj, cross(pr, wj));
        pv = pv + vJ;
        cJ = crm(pv, vJ);
    }
    B.bq = pq; B.r = pr; B.vel = pv; B.cJ = cJ;
}


struct WvCtx { // per-lane constants of the control step
    const float *s_patch;     // + lane
    int pox, poy;             // global cell index of patch cell (0,0)
    float cell_min, dt;
    DynTerrain Tp;            // the PATCH FRAME: origin = centre of patch cell (DYN_PATCH/2, DYN_PATCH/2), so cell_of(., Tp) is a patch index.
                              // Every horizontal coordinate of the contact code lives in this frame (|x| < 2 m, ulp 2.4e-7 m); the
                              // reference's env-local coordinates reach ~1 km at 65 536 envs (ulp 6e-5 m), see the kernel prologue.
};

// The wave's per-lane list of contact planes (parc_dynamics.hpp, contact manifold).  A plane is 8 floats (body-frame point, world normal,
// offset in the patch frame, weight), laid out [slot][field][lane]; slots [0, cap) live in LDS, the next WV_MAN_OVF in global memory (the
// lane's overflow area: only lanes with more planes than the wave's LDS share touch it), anything beyond is dropped and counted in
// g_wave_man_drops (parc_env_dynamics_manifold_drops: must stay 0, the GPU tests check it).  The list is filled body by body in the order
// the wave visits its bodies (`site` = the static index of the visit: records 0..2, limb bodies 3..5, wave 0's own trunk bodies 6..8) and
// walked in the same order in every substep: `cur` is the running start of the current body, `cnt` holds the per-site counts (6 bits each).
#define WV_MAN_OVF 20
struct WvMan { float *lds, *glb; int cap, cur; unsigned long long cnt; };
__device__ unsigned int g_wave_man_drops;

// a contact with a general normal: the shared statement of parc_dynamics.hpp
__device__ __forceinline__ void wv_contact_generic(const DynModel &M, float dt, v3 x, v3 vpt, float pen, v3 n, sym6 &IA, s6 &pA, v3 &fsum, float w) {
    contact_apply(M, dt, x, vpt, pen, n, IA, pA, fsum, w);
}

// Contacts at the point x (relative to O), summed in point space: force F and the symmetric 3x3 Bm (xx yy zz xy xz yz) of the implicit
// term.  pA -= [x x F; F];  IA += dt X^T Bm X with X = [-[x]x 1] (point acceleration = a_lin + alpha x x), i.e. the blocks
// [[-[x]x Bm [x]x, [x]x Bm], [., Bm]].  For Bm = beta 1 + (bn - beta) n n^T this is contact_apply's add_inertia + symrank1.
__device__ __forceinline__ void wv_point_apply(float dt, v3 x, v3 F, const float Bm_[6], sym6 &IA, s6 &pA, v3 &fsum) {
    const v3 no = cross(x, F);
    pA.a[0] -= no.x; pA.a[1] -= no.y; pA.a[2] -= no.z; pA.a[3] -= F.x; pA.a[4] -= F.y; pA.a[5] -= F.z;
    fsum = fsum + F;
    const float bxx = dt * Bm_[0], byy = dt * Bm_[1], bzz = dt * Bm_[2], bxy = dt * Bm_[3], bxz = dt * Bm_[4], byz = dt * Bm_[5];
    // C = [x]x Bm  (rows: -z B1 + y B2 | z B0 - x B2 | -y B0 + x B1, with Bk the k-th row of Bm)
    const float c00 = -x.z * bxy + x.y * bxz, c01 = -x.z * byy + x.y * byz, c02 = -x.z * byz + x.y * bzz;
    const float c10 = x.z * bxx - x.x * bxz, c11 = x.z * bxy - x.x * byz, c12 = x.z * bxz - x.x * bzz;
    const float c20 = -x.y * bxx + x.x * bxy, c21 = -x.y * bxy + x.x * byy, c22 = -x.y * bxz + x.x * byz;
    // upper-left block -C [x]x (symmetric): columns of [x]x are (0, z, -y), (-z, 0, x), (y, -x, 0)
    IA.s[sidx(0, 0)] += c02 * x.y - c01 * x.z; IA.s[sidx(0, 1)] += c00 * x.z - c02 * x.x; IA.s[sidx(0, 2)] += c01 * x.x - c00 * x.y;
    IA.s[sidx(1, 1)] += c10 * x.z - c12 * x.x; IA.s[sidx(1, 2)] += c11 * x.x - c10 * x.y;
    IA.s[sidx(2, 2)] += c21 * x.x - c20 * x.y;
    // upper-right block C
    IA.s[sidx(0, 3)] += c00; IA.s[sidx(0, 4)] += c01; IA.s[sidx(0, 5)] += c02;
    IA.s[sidx(1, 3)] += c10; IA.s[sidx(1, 4)] += c11; IA.s[sidx(1, 5)] += c12;
    IA.s[sidx(2, 3)] += c20; IA.s[sidx(2, 4)] += c21; IA.s[sidx(2, 5)] += c22;
    // lower-right block Bm
    IA.s[sidx(3, 3)] += bxx; IA.s[sidx(4, 4)] += byy; IA.s[sidx(5, 5)] += bzz; IA.s[sidx(3, 4)] += bxy; IA.s[sidx(3, 5)] += bxz; IA.s[sidx(4, 5)] += byz;
}

// articulated inertia / bias of body b: own inertia + contacts + (IA, pA) carried in from the chain's child
__device__ __forceinline__ void wv_body_inertia(const DynModel &M, const WaveTables &W, const DynTerrain &T, const WvCtx &X, int b, WvBody &B,
                                                const m3 &R, v3 rootp, sym6 &IA, s6 &pA, WvMan &man, const int site, const bool discover WSTAMP_PARAMS) {
    const v3 r = B.r;
    const float dt = X.dt;
    const DynTerrain &Tp = X.Tp;
    {
        const v3 cm = r + mulv(R, mk(W.c[b].com[0], W.c[b].com[1], W.c[b].com[2]));
        const float Ib[3][3] = {{W.c[b].inertia[0], W.c[b].inertia[3], W.c[b].inertia[4]}, {W.c[b].inertia[3], W.c[b].inertia[1], W.c[b].inertia[5]},
                                {W.c[b].inertia[4], W.c[b].inertia[5], W.c[b].inertia[2]}};
        float RI[3][3], Iw[3][3];
        PARC_UNROLL
        for (int a = 0; a < 3; ++a) {
            PARC_UNROLL
            for (int q = 0; q < 3; ++q) RI[a][q] = R.m[a][0] * Ib[0][q] + R.m[a][1] * Ib[1][q] + R.m[a][2] * Ib[2][q];
        }
        PARC_UNROLL
        for (int a = 0; a < 3; ++a) {
            PARC_UNROLL
            for (int q = 0; q < 3; ++q) Iw[a][q] = RI[a][0] * R.m[q][0] + RI[a][1] * R.m[q][1] + RI[a][2] * R.m[q][2];
        }
        const float Icw[6] = {Iw[0][0], Iw[1][1], Iw[2][2], Iw[0][1], Iw[0][2], Iw[1][2]};
        sym6 Iown;
        PARC_UNROLL
        for (int i = 0; i < 21; ++i) Iown.s[i] = 0.f;
        add_inertia(Iown, W.c[b].mass, cm, Icw);
        const s6 Iv = symmul(Iown, B.vel);
        const s6 pb = crf(B.vel, Iv);
        const v3 fg = mk(0.f, 0.f, W.c[b].mass * M.gravity_z);
        const v3 ng = cross(cm, fg);
        PARC_UNROLL
        for (int i = 0; i < 21; ++i) IA.s[i] += Iown.s[i];
        pA.a[0] += pb.a[0] - ng.x; pA.a[1] += pb.a[1] - ng.y; pA.a[2] += pb.a[2] - ng.z;
        pA.a[3] += pb.a[3] - fg.x; pA.a[4] += pb.a[4] - fg.y; pA.a[5] += pb.a[5] - fg.z;
    }
    WPIN(IA, pA);
    WSTAMP(12);
    // Contacts (parc_dynamics.hpp, contact manifold): DISCOVERY in the substeps 0, man_period, ... -- every candidate sphere of the body
    // against the columns it can reach within the speculative margin -> planes in this wave's per-lane list --, then, in EVERY substep, the
    // EVALUATION of the body's planes.  With lane = env a wave executes the narrow phase of a candidate as soon as ONE of its 64 envs needs
    // it (round 3: every candidate of every body, in every wave), so the discovery costs what the whole contact code cost in round 3; the
    // evaluation loop's trip count is the LARGEST per-lane plane count of the body in the wave (feet: 4-8 of 10 candidates x 4 columns).
    const int site6 = 6 * site;
    if (discover) {
        // Is the straight path applicable to this lane?  The sphere centres lie within brho of the bounding centre, i.e. (brho < one cell)
        // in the 3x3 cells around its cell, and each reaches at most the columns one cell further: with the bounding centre's cell at
        // least two cells inside the staged patch every height the narrow phase asks for is in LDS.  (Rounds 1-3 also kept a 5x5 running
        // maximum of the patch to cull bodies and points that clear every column of that window: with lane = env the cull never skipped
        // an instruction -- some lane of the wave always needs the candidate -- and its two table passes cost two block barriers in the
        // prologue; the narrow phase's own tests reject the same candidates.)
        const float brho = W.c[b].brho;
        const float smax = M.spec_max;
        bool inwin;
        {
            const v3 cb = r + mulv(R, mk(W.c[b].bc[0], W.c[b].bc[1], W.c[b].bc[2]));
            const int bx = cell_of(cb.x + rootp.x, Tp.min_x, Tp.dx), by = cell_of(cb.y + rootp.y, Tp.min_y, Tp.dy);
            inwin = brho < X.cell_min && bx >= 2 && bx < DYN_PATCH - 2 && by >= 2 && by < DYN_PATCH - 2;
        }
        // Narrow phase.  A sphere can only come within the margin of the column of its own cell and of the neighbours on the sides whose
        // face is closer than radius + margin (for rad + spec_max < half a cell: at most the x-side, the y-side and their diagonal; the far
        // sides are at least half a cell away); whether a candidate is emitted stays with sphere_vs_column / own_column_contact and the
        // margin test, so the result equals sphere_discover's exhaustive 9-column test (parc_dynamics.hpp).  Lanes with a point outside the
        // staged patch (or a sphere too wide for the near-side argument) take that exhaustive path below.
        const int npt = W.c[b].npt, pt0 = W.c[b].pt0, nsg = W.c[b].nsg, sg0 = W.c[b].sg0, own_npt = W.c[b].own_npt;
        const float hx = 0.5f * Tp.dx, hy = 0.5f * Tp.dy;
        bool slow = false;
        int n_new = 0; // planes this lane has pushed for the body
#ifdef PARC_COUNTS
        int cnt_pairs = 0, cnt_exec = 0; // wave-uniform
#endif
        // one plane into the lane's list: LDS slots [0, cap), then the overflow area in global memory, then dropped (counted)
        auto push = [&](v3 pb, v3 n, float off, float wq) __attribute__((always_inline)) {
            const int slot = man.cur + n_new;
            if (slot < man.cap) {
                float *e_ = man.lds + slot * (8 * 64);
                e_[0] = pb.x; e_[64] = pb.y; e_[128] = pb.z; e_[192] = n.x; e_[256] = n.y; e_[320] = n.z; e_[384] = off; e_[448] = wq;
                ++n_new;
            } else if (slot < man.cap + WV_MAN_OVF) {
                float *e_ = man.glb + (slot - man.cap) * (8 * 64);
                e_[0] = pb.x; e_[64] = pb.y; e_[128] = pb.z; e_[192] = n.x; e_[256] = n.y; e_[320] = n.z; e_[384] = off; e_[448] = wq;
                ++n_new;
            } else {
                atomicAdd(&g_wave_man_drops, 1u);
            }
        };
        auto seg_ends = [&](int ks, v3 &A, v3 &Bv) __attribute__((always_inline)) {
            A = r + mulv(R, mk(W.seg[ks][0], W.seg[ks][1], W.seg[ks][2])) + rootp;
            Bv = r + mulv(R, mk(W.seg[ks][4], W.seg[ks][5], W.seg[ks][6])) + rootp;
        };
        // edge candidate of segment ks (patch frame).  FAST: heights from the staged patch with clamped indices -- both ends of a segment
        // are collision points of the body, so a lane whose segment leaves the inner patch is a `slow` lane, which uses the other variant
        // (global memory outside the patch)
        auto seg_point = [&](int ks, v3 &Q) __attribute__((always_inline)) {
            v3 A, Bv;
            seg_ends(ks, A, Bv);
            return segment_edge_point(Tp, A, Bv, [&](int ix, int iy) {
                const int a_ = ix < 0 ? 0 : (ix > DYN_PATCH - 1 ? DYN_PATCH - 1 : ix), b_ = iy < 0 ? 0 : (iy > DYN_PATCH - 1 ? DYN_PATCH - 1 : iy);
                return X.s_patch[(a_ * DYN_PATCH + b_) * 64]; }, Q);
        };
        auto top_slow = [&](int ix, int iy) __attribute__((always_inline)) {
            return (ix >= 0 && ix < DYN_PATCH && iy >= 0 && iy < DYN_PATCH) ? X.s_patch[(ix * DYN_PATCH + iy) * 64] : hf_at(T, X.pox + ix, X.poy + iy); };
        auto seg_point_slow = [&](int ks, v3 &Q) __attribute__((always_inline)) {
            v3 A, Bv;
            seg_ends(ks, A, Bv);
            return segment_edge_point(Tp, A, Bv, top_slow, Q);
        };
        // Which lanes take the straight path: the body's 5x5 window is applicable (all its spheres then lie in the inner patch) and every
        // sphere + margin is narrower than half a cell.  The others take the exhaustive path further down.
        {
            const float fast_r = X.cell_min * 0.5f - 2e-3f - smax;
            bool wide = false;
            for (int pi = 0; pi < npt; ++pi) wide = wide || !(W.colp[pt0 + pi][3] < fast_r); // uniform
            slow = !inwin || wide;
        }
        // narrow phase of one candidate sphere (body-frame centre pb, x relative to O, g = x + rootp in the patch frame) of a lane whose
        // spheres all lie in the inner patch: own column, then the neighbour columns on the sides whose face is closer than radius + margin
        auto narrow = [&](v3 pb, v3 x, v3 g, float rad, float wq) __attribute__((always_inline)) {
            const int pa_ = cell_of(g.x, Tp.min_x, Tp.dx), pb_ = cell_of(g.y, Tp.min_y, Tp.dy); // patch indices = cell indices of the patch frame
            const float ex = g.x - (Tp.min_x + (float)pa_ * Tp.dx), ey = g.y - (Tp.min_y + (float)pb_ * Tp.dy);
            const int sx = ex >= 0.f ? 1 : -1, sy = ey >= 0.f ? 1 : -1;
            // every height this candidate can ask for -- own cell, the four face neighbours, the diagonal on the centre's side -- is requested
            // up front: six LDS reads in flight together instead of up to eight dependent round trips
            const float *hp = X.s_patch + (pa_ * DYN_PATCH + pb_) * 64;
            float top0 = hp[0], hxp = hp[DYN_PATCH * 64], hxm = hp[-DYN_PATCH * 64], hyp = hp[64], hym = hp[-64], hd = hp[(sx * DYN_PATCH + sy) * 64];
            asm volatile("" : "+v"(top0), "+v"(hxp), "+v"(hxm), "+v"(hyp), "+v"(hym), "+v"(hd));
            const v3 vpt = s6lin(B.vel) + cross(s6ang(B.vel), x);
            const float zlo = g.z - rad - spec_margin(M, vpt.z);
            if (!(zlo > top0)) { // own column: centre above the surface -> normal +z; centre inside the solid -> cheapest way out (own_column_contact)
                v3 n = mk(0.f, 0.f, 1.f);
                float pen = rad + top0 - g.z;
                if (g.z < top0) pen = own_column_contact(Tp, g, rad, pa_, pb_, top0, [&](int ox, int oy) { return ox == 1 ? hxp : (ox == -1 ? hxm : (oy == 1 ? hyp : hym)); }, n);
                if (pen > -spec_margin(M, dot(vpt, n))) push(pb, n, pen + dot(n, g), wq);
            }
            const float lim = rad + smax + 1e-3f; // slack >> the rounding of the cell centres: it only admits candidates
            const bool nx = hx - fabsf(ex) < lim, ny = hy - fabsf(ey) < lim;
            if (nx || ny) {
                for (int c = 0; c < 3; ++c) { // x side, y side, diagonal
                    const bool want = c == 0 ? nx : (c == 1 ? ny : (nx && ny));
                    if (!want) continue;
                    const int ox_ = c == 1 ? 0 : sx, oy_ = c == 0 ? 0 : sy;
                    const float top = c == 0 ? (sx > 0 ? hxp : hxm) : (c == 1 ? (sy > 0 ? hyp : hym) : hd);
                    if (!(top > top0 + 1e-3f) || zlo > top) continue; // only higher neighbours act as walls / step edges
                    v3 n;
                    const float pen = sphere_vs_column(Tp, g, rad, pa_ + ox_, pb_ + oy_, top, n);
                    if (pen > -spec_margin(M, dot(vpt, n))) push(pb, n, pen + dot(n, g), wq);
                }
            }
        };
        {   // the table entry of the NEXT point is requested while this one is processed (a scalar load waited for at the top of every
            // iteration is a ~200-cycle stall with one resident wave per SIMD)
            struct ColPt2 { float x, y, z, r; };
            auto load_pt2 = [&](int pi) __attribute__((always_inline)) {
                const int kp = pt0 + (pi < npt ? pi : 0);
                ColPt2 c; c.x = W.colp[kp][0]; c.y = W.colp[kp][1]; c.z = W.colp[kp][2]; c.r = W.colp[kp][3];
                return c;
            };
            ColPt2 cur = load_pt2(0);
            for (int pi = 0; pi < npt; ++pi) {
                const ColPt2 nxt = load_pt2(pi + 1);
                const v3 pb = mk(cur.x, cur.y, cur.z);
                const v3 x = r + mulv(R, pb);
                const v3 g = x + rootp;
                const bool mine = !slow;
#ifdef PARC_COUNTS
                cnt_pairs += __popcll(__ballot(mine)); cnt_exec += __any(mine) ? 1 : 0;
#endif
                if (__any(mine)) { // uniform
                    if (mine) narrow(pb, x, g, cur.r, pi < own_npt ? 1.f : -1.f); // a negative weight marks a plane of the merged fixed leaf (|w| acts)
                }
                cur = nxt;
            }
        }
        WPIN(IA, pA);
        WSTAMP(13); // narrow phase of the collision points
        // The body's segments: where a shaft / sole edge crosses a grid line with a step, the closest point to that edge is one more candidate
        // sphere (segment_edge_point, parc_dynamics.hpp), kept in the body frame like a point's centre.
        {
            for (int si = 0; si < nsg; ++si) {
                v3 Q = rootp;
                const float wq = seg_point(sg0 + si, Q);
                const bool has = wq > 0.f && !slow;
#ifdef PARC_COUNTS
                cnt_pairs += __popcll(__ballot(has)); cnt_exec += __any(has) ? 1 : 0;
#endif
                if (!__any(has)) continue; // uniform
                if (has) { const v3 x = Q - rootp; narrow(mulTv(R, x - r), x, Q, W.seg[sg0 + si][3], wq); }
            }
        }
        if (slow) { // exhaustive test of every candidate, heights from the patch where it covers them, else from global memory
            for (int pi = 0; pi < npt + nsg; ++pi) {
                v3 x, pb; float rad, wq = 1.f;
                if (pi < npt) {
                    const int kp = pt0 + pi;
                    pb = mk(W.colp[kp][0], W.colp[kp][1], W.colp[kp][2]);
                    x = r + mulv(R, pb);
                    rad = W.colp[kp][3];
                    if (pi >= own_npt) wq = -1.f;
                } else {
                    v3 Q = rootp;
                    wq = seg_point_slow(sg0 + pi - npt, Q);
                    if (!(wq > 0.f)) continue;
                    x = Q - rootp; rad = W.seg[sg0 + pi - npt][3];
                    pb = mulTv(R, x - r);
                }
                const v3 g = x + rootp;
                const v3 vpt = s6lin(B.vel) + cross(s6ang(B.vel), x);
                sphere_discover(M, Tp, g, rad, vpt, top_slow, [&](float pen, v3 n) { push(pb, n, pen + dot(n, g), wq); });
            }
        }
        man.cnt = (man.cnt & ~(63ull << site6)) | ((unsigned long long)n_new << site6);
#ifdef PARC_COUNTS
        {
            const unsigned long long m_near = __ballot(inwin), m_slow = __ballot(slow);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&g_wave_cnt[b][0], 64ull); atomicAdd(&g_wave_cnt[b][1], (unsigned long long)__popcll(m_near));
                atomicAdd(&g_wave_cnt[b][2], m_near ? 1ull : 0ull); atomicAdd(&g_wave_cnt[b][3], (unsigned long long)cnt_pairs);
                atomicAdd(&g_wave_cnt[b][4], (unsigned long long)(npt + nsg)); atomicAdd(&g_wave_cnt[b][5], (unsigned long long)__popcll(m_slow));
                atomicAdd(&g_wave_cnt[b][6], (unsigned long long)cnt_exec); atomicAdd(&g_wave_cnt[b][7], 1ull);
            }
        }
#endif
        WPIN(IA, pA);
        WSTAMP(6);  // segments (+ the exhaustive path of lanes outside the patch)
    }
    // ---- evaluation of the body's planes: x = r + R p, pen = off - n . (x + rootp); contact_apply on pen > 0 (the shared statement of
    // parc_dynamics.hpp).  Slots [cur, cur + n) of the lane's list; the trip count is the largest n in the wave.
    v3 fsum = mk(0.f, 0.f, 0.f), fsum_c = mk(0.f, 0.f, 0.f);
    {
        const int n_me = (int)((man.cnt >> site6) & 63ull);
        // plane t of the lane (valid address for every lane: slot 0 when it has no plane t)
        auto fetch = [&](int t, float (&e_)[8]) __attribute__((always_inline)) {
            const int slot = t < n_me ? man.cur + t : 0;
            const bool inl = slot < man.cap;
            const float *q_ = man.lds + (inl ? slot : 0) * (8 * 64);
            PARC_UNROLL
            for (int f = 0; f < 8; ++f) e_[f] = q_[f * 64];
            if (__any(!inl)) { // overflow area (rare): the lane's own earlier stores, global memory
                if (!inl) {
                    const float *g_ = man.glb + (slot - man.cap) * (8 * 64);
                    PARC_UNROLL
                    for (int f = 0; f < 8; ++f) e_[f] = __builtin_nontemporal_load(g_ + f * 64);
                }
            }
        };
        for (int t = 0; __any(t < n_me); ++t) {
            float e_[8];
            fetch(t, e_);
            const v3 x = r + mulv(R, mk(e_[0], e_[1], e_[2]));
            const v3 n = mk(e_[3], e_[4], e_[5]);
            const float pen = e_[6] - dot(n, x + rootp);
            const bool hit = t < n_me && pen > 0.f;
            if (__any(hit)) {
                const v3 vpt = s6lin(B.vel) + cross(s6ang(B.vel), x);
                // (tried in round 4, both slower by 3 %: a short form for normals that are exactly +z behind an __all test -- two code paths
                // instead of one --, and fetching plane t + 1 while plane t is evaluated -- eight more live registers and their copies; the latter
                // again +2.7 % after the legs had become the critical waves)
                // (tried in round 4, +0.6 %: the planes of a body partitioned after discovery, touching ones first, so that the late rounds
                // hold only speculative planes and skip the force law -- the rounds still find a penetrating lane, the partition costs)
                if (hit) {
                    v3 f1 = mk(0.f, 0.f, 0.f);
                    contact_apply(M, dt, x, vpt, pen, n, IA, pA, f1, fabsf(e_[7]));
                    fsum = fsum + f1;                                      // own + merged leaf; the leaf's share is taken out below
                    if (e_[7] < 0.f) fsum_c = fsum_c + f1;
                }
            }
        }
        man.cur += n_me;
    }
    B.fcon = fsum - fsum_c; B.fcon_c = fsum_c; // (the same addends in the same order: a body whose only contacts are its leaf's reports exactly 0)
    WPIN(IA, pA);
    WSTAMP(4);  // evaluation of the body's planes
}

// joint elimination of body b: (IA, pA) -> contribution (Ic, pc) to the parent; K and D^-1 u stay in `fac` (registers) for the outward pass
template <int FS, bool FF> // FS: stride of `fac` (1: a register array; 64: [slot][lane] in LDS); FF: see wv_load_joint
__device__ __forceinline__ void wv_joint_inward(const DynModel &M, const WaveTables &W, int b, const WvBody &B, const m3 &R, float dt, const sym6 &IA, const s6 &pA,
                                                sym6 &Ic, s6 &pc, float *fac) {
    const int jt = W.c[b].jtype, di = W.c[b].dof_idx;
    const v3 r = B.r;
    if (jt == DJ_SPHERICAL) {
        // The elimination of a spherical joint depends on the SUBSPACE its motion vectors span, not on their basis: S = P R with
        // P = [1; [r]x] (6 x 3) and R the body rotation, so span(S) = span(P) and the world-axis basis P serves -- U' = IA P is 36
        // multiply-adds instead of the 108 of IA S, D' = P^T U' + R diag(aug) R^T 39 instead of 54, and S itself is never formed
        // (round 4: -117 of ~560 vector instructions per elimination; parc_dynamics.hpp keeps the joint-frame statement, S = P R:
        // U = U' R, D = R^T D' R, K = K' R, K u = K' u', I - K U^T = I - K' U'^T -- the same projection up to rounding).  The factors kept
        // for the outward pass are K' and D'^-1 u' (world axes); the joint-frame acceleration the integrator needs is R^T q'.
        float tau[3], aug[3];
        const v3 err = FF ? mk(B.tq.x, B.tq.y, B.tq.z) : qlog(qmul(qconj(B.jq), B.tq)); // (FF: the feed-forward torque)
        const v3 cur = qlog(B.jq);
        const float e3[3] = {err.x, err.y, err.z}, c3[3] = {cur.x, cur.y, cur.z}, q3[3] = {B.qd.x, B.qd.y, B.qd.z};
        PARC_UNROLL
        for (int q = 0; q < 3; ++q) {
            float t, augq;
            if (!FF) {
                t = W.c[b].kp[q] * e3[q] - (W.c[b].kd[q] + dt * W.c[b].kp[q]) * q3[q];
                t = clampf(t, -W.c[b].eff[q], W.c[b].eff[q]);
                augq = W.c[b].arm[q] + dt * W.c[b].kd[q] + dt * dt * W.c[b].kp[q];
            } else drive_ff(M.ctrl, W.c[b].kd[q], W.c[b].arm[q], W.c[b].eff[q], dt, e3[q], q3[q], t, augq);
            aug[q] = augq;
            if (c3[q] < W.c[b].lo[q]) { t += M.lim_k * (W.c[b].lo[q] - c3[q]) - M.lim_d * q3[q]; aug[q] += dt * M.lim_d + dt * dt * M.lim_k; }
            else if (c3[q] > W.c[b].hi[q]) { t += M.lim_k * (W.c[b].hi[q] - c3[q]) - M.lim_d * q3[q]; aug[q] += dt * M.lim_d + dt * dt * M.lim_k; }
            tau[q] = t;
        }
        // U' = IA P: column j = IA[:, j] + sum_k IA[:, 3 + k] [r]x[k][j],  [r]x = [[0, -rz, ry], [rz, 0, -rx], [-ry, rx, 0]]
        s6 Uc[3];
        PARC_UNROLL
        for (int a = 0; a < 6; ++a) {
            const float i3 = sget(IA, a, 3), i4 = sget(IA, a, 4), i5 = sget(IA, a, 5);
            Uc[0].a[a] = sget(IA, a, 0) + (r.z * i4 - r.y * i5);
            Uc[1].a[a] = sget(IA, a, 1) + (r.x * i5 - r.z * i3);
            Uc[2].a[a] = sget(IA, a, 2) + (r.y * i3 - r.x * i4);
        }
        // u' = R tau - P^T pA,  P^T x = x_ang - [r]x x_lin = x_ang - r x x_lin
        const v3 tw = mulv(R, mk(tau[0], tau[1], tau[2]));
        const v3 pl = mk(pA.a[3], pA.a[4], pA.a[5]);
        const v3 rxp = cross(r, pl);
        const float uu[3] = {tw.x - (pA.a[0] - rxp.x), tw.y - (pA.a[1] - rxp.y), tw.z - (pA.a[2] - rxp.z)};
        // D' = P^T U' + R diag(aug) R^T (symmetric: six entries)
        float Dm[3][3];
        PARC_UNROLL
        for (int l = 0; l < 3; ++l) {
            const v3 ub = mk(Uc[l].a[3], Uc[l].a[4], Uc[l].a[5]);
            const v3 rxu = cross(r, ub);
            Dm[0][l] = Uc[l].a[0] - rxu.x; Dm[1][l] = Uc[l].a[1] - rxu.y; Dm[2][l] = Uc[l].a[2] - rxu.z;
        }
        PARC_UNROLL
        for (int q = 0; q < 3; ++q) {
            PARC_UNROLL
            for (int l = q; l < 3; ++l) {
                const float g = R.m[q][0] * aug[0] * R.m[l][0] + R.m[q][1] * aug[1] * R.m[l][1] + R.m[q][2] * aug[2] * R.m[l][2];
                Dm[q][l] += g;
                if (l != q) Dm[l][q] = Dm[q][l]; // the exact matrix is symmetric: one triangle decides
            }
        }
        const float c00 = Dm[1][1] * Dm[2][2] - Dm[1][2] * Dm[2][1], c01 = Dm[0][2] * Dm[2][1] - Dm[0][1] * Dm[2][2],
                    c02 = Dm[0][1] * Dm[1][2] - Dm[0][2] * Dm[1][1];
        const float id = DYN_RCP(Dm[0][0] * c00 + Dm[1][0] * c01 + Dm[2][0] * c02);
        float D3[3][3];
        D3[0][0] = c00 * id; D3[0][1] = c01 * id; D3[0][2] = c02 * id;
        D3[1][1] = (Dm[0][0] * Dm[2][2] - Dm[0][2] * Dm[2][0]) * id;
        D3[1][2] = (Dm[0][2] * Dm[1][0] - Dm[0][0] * Dm[1][2]) * id;
        D3[2][2] = (Dm[0][0] * Dm[1][1] - Dm[0][1] * Dm[1][0]) * id;
        D3[1][0] = D3[0][1]; D3[2][0] = D3[0][2]; D3[2][1] = D3[1][2];
        s6 Kc[3];
        float du[3];
        PARC_UNROLL
        for (int q = 0; q < 3; ++q) {
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) Kc[q].a[a] = Uc[0].a[a] * D3[0][q] + Uc[1].a[a] * D3[1][q] + Uc[2].a[a] * D3[2][q];
            du[q] = D3[q][0] * uu[0] + D3[q][1] * uu[1] + D3[q][2] * uu[2];
        }
        Ic = IA;
        PARC_UNROLL
        for (int a = 0; a < 6; ++a) {
            PARC_UNROLL
            for (int q = a; q < 6; ++q) Ic.s[sidx(a, q)] -= Kc[0].a[a] * Uc[0].a[q] + Kc[1].a[a] * Uc[1].a[q] + Kc[2].a[a] * Uc[2].a[q];
        }
        const s6 Iac = symmul(Ic, B.cJ);
        PARC_UNROLL
        for (int a = 0; a < 6; ++a) pc.a[a] = pA.a[a] + Iac.a[a] + Kc[0].a[a] * uu[0] + Kc[1].a[a] * uu[1] + Kc[2].a[a] * uu[2];
        PARC_UNROLL
        for (int q = 0; q < 3; ++q) {
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) fac[(6 * q + a) * FS] = Kc[q].a[a];
            fac[(18 + q) * FS] = du[q];
        }
    } else if (jt == DJ_HINGE) {
        const v3 a = mulv(R, mk(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]));
        const s6 Sc = s6mk(a, cross(r, a));
        float t, aug;
        if (!FF) {
            t = W.c[b].kp[0] * (B.thang - B.hang) - (W.c[b].kd[0] + dt * W.c[b].kp[0]) * B.qd.x;
            t = clampf(t, -W.c[b].eff[0], W.c[b].eff[0]);
            aug = W.c[b].arm[0] + dt * W.c[b].kd[0] + dt * dt * W.c[b].kp[0];
        } else drive_ff(M.ctrl, W.c[b].kd[0], W.c[b].arm[0], W.c[b].eff[0], dt, B.thang, B.qd.x, t, aug);
        if (B.hang < W.c[b].lo[0]) { t += M.lim_k * (W.c[b].lo[0] - B.hang) - M.lim_d * B.qd.x; aug += dt * M.lim_d + dt * dt * M.lim_k; }
        else if (B.hang > W.c[b].hi[0]) { t += M.lim_k * (W.c[b].hi[0] - B.hang) - M.lim_d * B.qd.x; aug += dt * M.lim_d + dt * dt * M.lim_k; }
        const s6 Uc = symmul(IA, Sc);
        float sp = 0.f, d = aug;
        PARC_UNROLL
        for (int q = 0; q < 6; ++q) { sp += Sc.a[q] * pA.a[q]; d += Sc.a[q] * Uc.a[q]; }
        const float uu = t - sp, di_ = DYN_RCP(d);
        s6 Kc;
        PARC_UNROLL
        for (int q = 0; q < 6; ++q) Kc.a[q] = Uc.a[q] * di_;
        Ic = IA;
        PARC_UNROLL
        for (int q = 0; q < 6; ++q) {
            PARC_UNROLL
            for (int l = q; l < 6; ++l) Ic.s[sidx(q, l)] -= Kc.a[q] * Uc.a[l];
        }
        const s6 Iac = symmul(Ic, B.cJ);
        PARC_UNROLL
        for (int q = 0; q < 6; ++q) { pc.a[q] = pA.a[q] + Iac.a[q] + Kc.a[q] * uu; fac[q * FS] = Kc.a[q]; }
        fac[6 * FS] = di_ * uu;
    } else { // fixed joint
        const s6 Iac = symmul(IA, B.cJ);
        Ic = IA;
        pc = pA + Iac;
    }
}

// acceleration of body b from its parent's (ap); leaves its own in ap
template <int FS>
__device__ __forceinline__ void wv_joint_outward(const DynModel &M, const WaveTables &W, int b, WvBody &B, s6 &ap, const float *fac) {
    const int jt = W.c[b].jtype;
    s6 ai = ap + B.cJ;
    if (jt == DJ_SPHERICAL) {
        float q3[3];
        PARC_UNROLL
        for (int q = 0; q < 3; ++q) {
            float ka = 0.f;
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) ka += fac[(6 * q + a) * FS] * ai.a[a];
            q3[q] = fac[(18 + q) * FS] - ka;
        }
        const v3 wj = mk(q3[0], q3[1], q3[2]);       // the joint's angular acceleration in WORLD axes (the factors are K', D'^-1 u': wv_joint_inward)
        B.qdd = mulTv(qmat(B.bq), wj);               // ... and in the joint (child) frame, which the integrator advances
        ai = ai + s6mk(wj, cross(B.r, wj));
    } else if (jt == DJ_HINGE) {
        float ka = 0.f;
        PARC_UNROLL
        for (int a = 0; a < 6; ++a) ka += fac[a * FS] * ai.a[a];
        const float qa = fac[6 * FS] - ka;
        B.qdd = mk(qa, 0.f, 0.f);
        const v3 wj = qa * mulv(qmat(B.bq), mk(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]));
        ai = ai + s6mk(wj, cross(B.r, wj));
    }
    ap = ai;
}

__device__ __forceinline__ void wv_integrate_joint(const DynModel &M, const WaveTables &W, int b, WvBody &B, float dt) {
    const int jt = W.c[b].jtype;
    const float mw = M.max_ang_vel;
    if (jt == DJ_SPHERICAL) {
        B.qd = mk(clampf(B.qd.x + dt * B.qdd.x, -mw, mw), clampf(B.qd.y + dt * B.qdd.y, -mw, mw), clampf(B.qd.z + dt * B.qdd.z, -mw, mw));
        B.jq = qnormalize(qmul(B.jq, qexp(dt * B.qd)));
    } else if (jt == DJ_HINGE) {
        B.qd.x = clampf(B.qd.x + dt * B.qdd.x, -mw, mw);
        B.hang += dt * B.qd.x;
    }
}

// pr (optional): the env's prep record for the observation kernel (k_env_prep's output, parc_env.hip): slot b = dof -> quat of joint b
// as the reference's kinematic model forms it FROM THE STORED DOFS (kin_char_model.py:586, torch_util.py:70-91, :337) -- the same
// device functions on the same fp32 inputs as k_env_prep, so the record is bit-identical and that launch is not needed after a step.
__device__ __forceinline__ void wv_store_joint(const DynModel &M, const WaveTables &W, int b, const WvBody &B, float *dp, float *dv, float *cf, float4 *pr) {
    const int jt = W.c[b].jtype, di = W.c[b].dof_idx;
    // plain copies first: selecting between struct fields inside the branches would keep the body in scratch
    const q4 jq = B.jq; const float hang = B.hang; const v3 qd = B.qd, fc = B.fcon;
    if (jt == DJ_SPHERICAL) {
        const v3 ex = qlog(jq);
        dp[di] = ex.x; dp[di + 1] = ex.y; dp[di + 2] = ex.z; dv[di] = qd.x; dv[di + 1] = qd.y; dv[di + 2] = qd.z;
        if (pr) pr[b] = parc::exp_map_to_quat(parc::mk3(ex.x, ex.y, ex.z));
    } else if (jt == DJ_HINGE) {
        dp[di] = hang; dv[di] = qd.x;
        if (pr) pr[b] = parc::axis_angle_to_quat(parc::mk3(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]), hang);
    } else if (pr && b != 0) { // fixed joint (slot 0, the root body's, holds the heading terms)
        pr[b] = make_float4(0.f, 0.f, 0.f, 1.f);
    }
    cf[3 * b] = fc.x; cf[3 * b + 1] = fc.y; cf[3 * b + 2] = fc.z;
    const int ch = W.c[b].child; // the fixed leaf merged into this body: its contact force, its (identity) prep-record quaternion
    if (ch >= 0) {
        const v3 fcc = B.fcon_c;
        cf[3 * ch] = fcc.x; cf[3 * ch + 1] = fcc.y; cf[3 * ch + 2] = fcc.z;
        if (pr) pr[ch] = make_float4(0.f, 0.f, 0.f, 1.f);
    }
}

template <bool FF> // the kernel's body; FF = a control mode other than pd (k_dynamics_wave_ff): an instantiation of its own, the default one carries none of it
__device__ __forceinline__ void dynamics_wave_body(const DynModel *__restrict__ Mp, const WaveTables *__restrict__ Wp, DynTerrain T,
                                                   ParcEnvBuffers buf, const float *__restrict__ action,
                                                   const float *__restrict__ env_off_all, float *__restrict__ root_shadow,
                                                   float4 *__restrict__ prep, float *__restrict__ man_g, int N, int epb) {
    extern __shared__ float smem[];
    const DynModel &M = *Mp;
    const WaveTables &W = *Wp;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int e = blockIdx.x * epb + lane; // epb = envs per block: 64, or 32 (half-filled waves) when that is what it takes to put a block on every CU
    const bool env_ok = e < N;
    const int ec = env_ok ? e : N - 1; // clamp for address safety; lanes past N compute on a copy and store nothing
    const float dt = M.dt;
    const int B_ = M.B, D_ = M.D, nsub = M.nsub;
    const int lc = w + 1;              // this wave's limb chain
#ifdef PARC_STAMPS
    unsigned long long wacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_readcyclecounter();
#endif
    const bool has_limb = w < W.nlimb;
    const int llen = has_limb ? W.len[lc] : 0, tlen = w == 0 ? W.len[0] : 0;

    float *s_attkin = smem + WV_OFF_ATTKIN + lane, *s_up = smem + WV_OFF_UP + lane, *s_attacc = smem + WV_OFF_ATTACC + lane;
    float *s_rootp = smem + WV_OFF_ROOTP + lane, *s_patch = smem + WV_OFF_PATCH + lane;
    float *s_rooti = smem + WV_OFF_ROOTI + lane;
    // hand-off flags: workgroup-scope atomics in LDS (release store by the producer after its record, acquire load by the consumer before
    // it reads the record); [15] = "a wait of this block timed out"
    int *s_flag = reinterpret_cast<int *>(smem + WV_OFF_FLAG);
    if (threadIdx.x < 16) s_flag[threadIdx.x] = 0; // published before the first barrier below
    if (lane >= epb) return; // lanes without an env leave (the wave goes on with the others; barriers and flags are per wave)

    const float *dp = buf.char_dof_pos + (size_t)D_ * ec, *dv = buf.char_dof_vel + (size_t)D_ * ec, *ac = action + (size_t)D_ * ec;
    WvBody limb[WV_MAXLEN], trunk[WV_MAXLEN];
    // joint-space factors K = U D^-1, D^-1 u between the inward and the outward pass: the limb's in registers; wave 0's trunk factors
    // (positions 1.., the root has none) in LDS -- with them in registers too the kernel spills (512 registers + 76 B of scratch per lane)
    float fac_l[WV_MAXLEN][WV_FAC];
#ifdef PARC_TRUNK_FAC_REGS
    float fac_t[WV_MAXLEN][WV_FAC];
#define WV_TFS 1
#define WV_TFAC(k) fac_t[k]
#else
    float *s_tfac = smem + WV_OFF_TFAC + lane;
#define WV_TFS 64
#define WV_TFAC(k) (s_tfac + ((k) - 1) * WV_FAC * 64)
#endif
    WvMan man;                                                // this wave's per-lane list of contact planes
    man.lds = smem + WV_OFF_MAN + W.man_base[w] * (8 * 64) + lane; man.cap = W.man_cap[w];
    man.glb = man_g + ((size_t)blockIdx.x * WV_MAXLIMB + w) * (WV_MAN_OVF * 8 * 64) + lane;
    man.cur = 0; man.cnt = 0ull;
    const v3 rp_buf = mk(buf.char_root_pos[3 * ec], buf.char_root_pos[3 * ec + 1], buf.char_root_pos[3 * ec + 2]);
    q4 rq; rq.x = buf.char_root_rot[4 * ec]; rq.y = buf.char_root_rot[4 * ec + 1]; rq.z = buf.char_root_rot[4 * ec + 2]; rq.w = buf.char_root_rot[4 * ec + 3];
    rq = qnormalize(rq);
    v3 rv = mk(buf.char_root_vel[3 * ec], buf.char_root_vel[3 * ec + 1], buf.char_root_vel[3 * ec + 2]);
    v3 rw = mk(buf.char_root_ang_vel[3 * ec], buf.char_root_ang_vel[3 * ec + 1], buf.char_root_ang_vel[3 * ec + 2]);

    WvCtx X;
    X.s_patch = s_patch; X.dt = dt;

    const float eo0 = env_off_all[3 * ec], eo1 = env_off_all[3 * ec + 1], eo2 = env_off_all[3 * ec + 2];
    X.cell_min = fminf(T.dx, T.dy);
    X.pox = cell_of(rp_buf.x + eo0, T.min_x, T.dx) - DYN_PATCH / 2; X.poy = cell_of(rp_buf.y + eo1, T.min_y, T.dy) - DYN_PATCH / 2;
    X.Tp = T; X.Tp.hf = nullptr; X.Tp.min_x = -(float)(DYN_PATCH / 2) * T.dx; X.Tp.min_y = -(float)(DYN_PATCH / 2) * T.dy;
    // PRECISION.  The state buffers hold the root in the reference's env-local frame (ig_parkour_env.py:386-398), whose
    // coordinates reach ~1 km at 65 536 envs: one ulp is 6e-5 m there, so `x += dt v` at dt = 1/120 s would drop every
    // |v| < 7 mm/s of a far env.  The step therefore runs in the patch frame, whose origin is the centre of the root's cell:
    //   anc   = that centre in env-local coordinates (any nearby representable value serves: it is fixed for the step);
    //   rp    = (buffer - anc) + residual: the subtraction of two neighbours is exact; the residual is what the last
    //           step's write-back rounded away (kept in library memory, valid while the buffer still holds what that step
    //           wrote, i.e. unless the caller reset or edited the state);
    //   out   = fl(anc + rp) goes back to the buffer, rp - (out - anc) becomes the new residual.
    // The observation / reward path keeps reading the reference-format buffer (bit-for-bit the reference's quantisation).
    const v3 anc = mk((T.min_x + (float)(X.pox + DYN_PATCH / 2) * T.dx) - eo0, (T.min_y + (float)(X.poy + DYN_PATCH / 2) * T.dy) - eo1, -eo2);
    v3 rlo = mk(0.f, 0.f, 0.f);
    if (root_shadow) {
        const float *sh = root_shadow + 6 * (size_t)ec;
        if (sh[0] == rp_buf.x && sh[1] == rp_buf.y && sh[2] == rp_buf.z) rlo = mk(sh[3], sh[4], sh[5]);
    }
    v3 rp = mk((rp_buf.x - anc.x) + rlo.x, (rp_buf.y - anc.y) + rlo.y, (rp_buf.z - anc.z) + rlo.z);
    // local height patch of each env (the 4 waves share the 81 cells).  The loops are
    // fully unrolled with a uniform predicate so that all loads of a pass are in flight together (one resident wave per
    // SIMD: a load per iteration would expose its whole latency 20 times).
    {
        constexpr int NP = (DYN_PATCH * DYN_PATCH + 3) / 4;
        float hv[NP];
        PARC_UNROLL
        for (int it = 0; it < NP; ++it) {
            const int i = w + 4 * it;
            hv[it] = i < DYN_PATCH * DYN_PATCH ? hf_at(T, X.pox + i / DYN_PATCH, X.poy + i % DYN_PATCH) : 0.f;
        }
        // the joint state (loads, then two exp maps per spherical joint) in the shadow of the patch loads, which sit at the end of the
        // prologue's only dependent chain of global loads (root position + env origin -> cell -> heights)
        PARC_UNROLL
        for (int k = 0; k < WV_MAXLEN; ++k) {
            if (k < llen) wv_load_joint<FF>(M, W, W.body[lc][k], limb[k], dp, dv, ac);
            if (k < tlen) wv_load_joint<FF>(M, W, W.body[0][k], trunk[k], dp, dv, ac);
        }
        PARC_UNROLL
        for (int it = 0; it < NP; ++it) {
            const int i = w + 4 * it;
            if (i < DYN_PATCH * DYN_PATCH) s_patch[i * 64] = hv[it];
        }
    }
    __syncthreads();

    WSTAMP(0); // prologue: state load, height patch
    // ---- substeps.  The four waves of a block meet at NO barrier inside the loop: every hand-off is a record in LDS plus a
    // flag (the producer writes the record, then the flag: LDS operations of one wave complete in order; the consumer polls
    // the flag, then reads).  A wave waits only for what it really depends on:
    //   limbs:  attach kinematics (wave 0)  ->  kinematics, inward pass, hand-over  ->  attach acceleration (wave 0)  ->  outward pass
    //   wave 0: trunk kinematics -> its own limb -> upper trunk (waits for the limbs hanging there and for the helper's records)
    //           -> root solve (waits for the limbs hanging off the root) -> trunk outward -> NEXT substep's trunk kinematics
    //           -> its own limb's outward pass (moved behind the next kinematics: the other limbs wait for that, not for this)
    // Re-use of a record by the next substep is ordered by the same chain (e.g. a limb overwrites its hand-over only after
    // it has received the acceleration that wave 0 computed from the previous one).
    auto publish = [&](int f, int sub) __attribute__((always_inline)) {
#ifdef PARC_TEST_BREAK_FLAG
        if (f == (PARC_TEST_BREAK_FLAG)) return;
#endif
        // every lane wrote its env's slot of the record: the wave's LDS stores complete in order, the fence waits for them (one
        // s_waitcnt for the whole wave), then lane 0 raises the flag
        // (the fences of publish / await order LDS traffic only -- every hand-off record lives in LDS --: the "local" form does not wait for
        // the wave's global stores to drain, which in the epilogue are ~100 row-strided stores per lane, 18 k cycles)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) __hip_atomic_store(&s_flag[f], sub + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto await = [&](int f, int sub) __attribute__((always_inline)) {
        int spins = 0;
        // bounded: a protocol error must not hang the GPU, and must not go unnoticed: the block's envs are poisoned at the end (s_flag[15])
        // and the device counter is bumped here, in the cold path only
        while (__hip_atomic_load(&s_flag[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= sub) {
            // once a wait of this block has timed out the step is void (the epilogue poisons the block's envs): the waits that follow give
            // up after 256 polls instead of spinning the full bound again -- a protocol error costs one bound per block, not one per wait
            if ((++spins & 255) == 0 && spins != PARC_FLAG_SPIN_BOUND && __hip_atomic_load(&s_flag[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
            if (spins == PARC_FLAG_SPIN_BOUND) {
                __hip_atomic_store(&s_flag[15], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0) atomicAdd(&g_wave_timeouts, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    auto trunk_kinematics = [&](int sub) __attribute__((always_inline)) { // wave 0
        s_rootp[0] = rp.x; s_rootp[64] = rp.y; s_rootp[128] = rp.z;
        q4 pq = rq; v3 pr = mk(0.f, 0.f, 0.f); s6 pv = s6mk(rw, rv);
        PARC_UNROLL
        for (int k = 0; k < WV_MAXLEN; ++k) {
            if (k < tlen) {
                wv_fk_body(M, W, W.body[0][k], trunk[k], pq, pr, pv);
                const int slot = W.att_slot[k];
                if (slot >= 0) {
                    float *s = s_attkin + slot * 13 * 64;
                    s[0] = pq.x; s[64] = pq.y; s[128] = pq.z; s[192] = pq.w; s[256] = pr.x; s[320] = pr.y; s[384] = pr.z;
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) s[(7 + a) * 64] = pv.a[a];
                    publish(WV_F_KIN(slot), sub);
                }
            }
        }
    };
    if (w == 0) trunk_kinematics(0);
    WSTAMP(1);
    for (int sub = 0; sub < nsub; ++sub) {
        if (w != 0) { // the root position travels with the root body's slot; a limb also needs its own parent's slot
            if (W.att_slot[0] >= 0) await(WV_F_KIN(W.att_slot[0]), sub);
            if (has_limb) await(WV_F_KIN(W.par_slot[lc]), sub);
        }
        WSTAMP(2);
        WTL(0); // substep start: parent kinematics received (limbs) / own trunk kinematics done (wave 0)
        const v3 rootp = mk(s_rootp[0], s_rootp[64], s_rootp[128]);
        const bool discover = sub % M.man_period == 0; // contact discovery in this substep (uniform); the others re-evaluate the cached planes
        man.cur = 0;
        // ---- inward pass of this wave's limb -----------------------------------------------------------------------------
        sym6 Icl; s6 pcl = s6zero();   // carry of this wave's limb
        PARC_UNROLL
        for (int i = 0; i < 21; ++i) Icl.s[i] = 0.f;
        const bool early = has_limb && W.early[lc] != 0;
        auto limb_body = [&](int k) __attribute__((always_inline)) {
            const int b = W.body[lc][k];
            const m3 R = qmat(limb[k].bq);
            sym6 IA = Icl; s6 pA = pcl;
            WSTAMP(15);
            wv_body_inertia(M, W, T, X, b, limb[k], R, rootp, IA, pA, man, 3 + k, discover WSTAMP_ARGS);
            wv_joint_inward<1, FF>(M, W, b, limb[k], R, dt, IA, pA, Icl, pcl, fac_l[k]);
            WPIN(Icl, pcl);
            WSTAMP(14);
        };
        // own inertia + contacts of a trunk body on behalf of wave 0, from the body's kinematics in its attach slot; the result travels as a
        // record + flag (rec_wave in build_wave_tables says who does which)
        auto prep_record = [&](int k) __attribute__((always_inline)) {
            await(WV_F_KIN(W.att_slot[k]), sub);
            const float *s = s_attkin + W.att_slot[k] * 13 * 64;
            WvBody rb;
            rb.bq.x = s[0]; rb.bq.y = s[64]; rb.bq.z = s[128]; rb.bq.w = s[192];
            rb.r = mk(s[256], s[320], s[384]);
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) rb.vel.a[a] = s[(7 + a) * 64];
            sym6 IA; s6 pA = s6zero();
            PARC_UNROLL
            for (int i = 0; i < 21; ++i) IA.s[i] = 0.f;
            WSTAMP(15);
            wv_body_inertia(M, W, T, X, W.body[0][k], rb, qmat(rb.bq), rootp, IA, pA, man, k, discover WSTAMP_ARGS);
            float *sr = s_rooti + k * 30 * 64;
            PARC_UNROLL
            for (int i = 0; i < 21; ++i) sr[i * 64] = IA.s[i];
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) sr[(21 + a) * 64] = pA.a[a];
            sr[27 * 64] = rb.fcon.x; sr[28 * 64] = rb.fcon.y; sr[29 * 64] = rb.fcon.z;
            publish(WV_F_REC(k), sub);
            WTL(3 + k); // record of trunk position k published (5: head, 4: torso, 3: pelvis)
        };
        if (has_limb) {
            const float *s = s_attkin + W.par_slot[lc] * 13 * 64;
            q4 pq; pq.x = s[0]; pq.y = s[64]; pq.z = s[128]; pq.w = s[192];
            v3 pr = mk(s[256], s[320], s[384]);
            s6 pv;
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) pv.a[a] = s[(7 + a) * 64];
            PARC_UNROLL
            for (int k = 0; k < WV_MAXLEN; ++k) if (k < llen) wv_fk_body(M, W, W.body[lc][k], limb[k], pq, pr, pv);
            WSTAMP(11);
            WTL(1); // limb kinematics done
            // records of the upper trunk bodies this wave prepares: wave 0 needs them before it needs this wave's limb
            PARC_UNROLL
            for (int k = WV_MAXLEN - 1; k >= 1; --k) if (w != 0 && W.rec_wave[k] == w) prep_record(k);
            PARC_UNROLL
            for (int k = WV_MAXLEN - 1; k >= 0; --k) if (k < llen) limb_body(k);
            float *u = s_up + (lc - 1) * 27 * 64;
            PARC_UNROLL
            for (int i = 0; i < 21; ++i) u[i * 64] = Icl.s[i];
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) u[(21 + a) * 64] = pcl.a[a];
            if (w != 0) publish(WV_F_UP(lc), sub); // wave 0 consumes its own limb's record in program order
            WTL(2); // limb inward pass done, hand-over published
        }
        if (W.rec_wave[0] == w) prep_record(0); // the root body's record: needed last, so after this wave's own limb
        WSTAMP(3);
#ifdef PARC_COUNTS
        if (sub % M.man_period == 0) { // planes this wave holds after a discovery: histogram of the largest per-lane count
            int mb = man.cur;
            for (int o = 32; o > 0; o >>= 1) mb = max(mb, __shfl_xor(mb, o));
            if (lane == 0) atomicAdd(&g_wave_hist[15][w][min(mb, 15)], 1ull);
        }
#endif
        // ---- wave 0: trunk inward pass, floating-base solve, trunk outward pass, integration -----------------------------
        if (w == 0) {
            sym6 Ict; s6 pct = s6zero();   // carry of the trunk
            PARC_UNROLL
            for (int i = 0; i < 21; ++i) Ict.s[i] = 0.f;
            s6 acc_root = s6zero();
            auto trunk_body = [&](int k) __attribute__((always_inline)) {
                const int b = W.body[0][k];
                const m3 R = qmat(trunk[k].bq);
                sym6 IA = Ict; s6 pA = pct;
                WTL(3 + 3 * k); // wave 0: starts trunk position k (9: head, 6: torso, 3: pelvis)
                if (W.prep[k]) { // prepared by the helper wave
                    await(WV_F_REC(k), sub);
                    WTL(4 + 3 * k); // its record arrived
                    const float *sr = s_rooti + k * 30 * 64;
                    PARC_UNROLL
                    for (int i = 0; i < 21; ++i) IA.s[i] += sr[i * 64];
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) pA.a[a] += sr[(21 + a) * 64];
                    trunk[k].fcon = mk(sr[27 * 64], sr[28 * 64], sr[29 * 64]);
                } else {
                    WSTAMP(15);
                    wv_body_inertia(M, W, T, X, b, trunk[k], R, rootp, IA, pA, man, 6 + k, discover WSTAMP_ARGS);
                }
                for (int ci = 0; ci < W.nchild[k]; ++ci) {
                    const int c = W.child[k][ci];
                    if (c != lc) await(WV_F_UP(c), sub);
                    const float *u = s_up + (c - 1) * 27 * 64;
                    PARC_UNROLL
                    for (int i = 0; i < 21; ++i) IA.s[i] += u[i * 64];
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) pA.a[a] += u[(21 + a) * 64];
                }
                WTL(5 + 3 * k); // all children of position k arrived and added
                if (b == 0) { // floating base: solve IA a0 = -pA (Cholesky)
                    float Lm[6][6];
                    PARC_UNROLL
                    for (int j = 0; j < 6; ++j) {
                        float sd = sget(IA, j, j);
                        PARC_UNROLL
                        for (int q = 0; q < j; ++q) sd -= Lm[j][q] * Lm[j][q];
                        sd = sd > 1e-12f ? DYN_SQRT(sd) : 1e3f; // a non-positive pivot is a numerical breakdown: treat the direction as immovable rather than as massless
                        const float isd = DYN_RCP(sd);
                        PARC_UNROLL
                        for (int a = j + 1; a < 6; ++a) {
                            float sa = sget(IA, a, j);
                            PARC_UNROLL
                            for (int q = 0; q < j; ++q) sa -= Lm[a][q] * Lm[j][q];
                            Lm[a][j] = sa * isd;
                        }
                        Lm[j][j] = isd; // the reciprocal of the pivot
                    }
                    float y[6], xs[6];
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) {
                        float sa = -pA.a[a];
                        PARC_UNROLL
                        for (int q = 0; q < a; ++q) sa -= Lm[a][q] * y[q];
                        y[a] = sa * Lm[a][a];
                    }
                    PARC_UNROLL
                    for (int a_ = 0; a_ < 6; ++a_) {
                        const int a = 5 - a_;
                        float sa = y[a];
                        PARC_UNROLL
                        for (int q = a + 1; q < 6; ++q) sa -= Lm[q][a] * xs[q];
                        xs[a] = sa * Lm[a][a];
                    }
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) acc_root.a[a] = xs[a];
                } else {
                    wv_joint_inward<WV_TFS, FF>(M, W, b, trunk[k], R, dt, IA, pA, Ict, pct, WV_TFAC(k));
                }
            };
            PARC_UNROLL
            for (int k = WV_MAXLEN - 1; k >= 0; --k) if (k < tlen) trunk_body(k);
            WSTAMP(5);
            s6 ap = acc_root;
            PARC_UNROLL
            for (int k = 0; k < WV_MAXLEN; ++k) {
                if (k < tlen) {
                    const int b = W.body[0][k];
                    if (b != 0) wv_joint_outward<WV_TFS>(M, W, b, trunk[k], ap, WV_TFAC(k));
                    const int slot = W.att_slot[k];
                    if (slot >= 0) { // the limbs hanging here can start their outward pass
                        PARC_UNROLL
                        for (int a = 0; a < 6; ++a) s_attacc[(slot * 6 + a) * 64] = ap.a[a];
                        publish(WV_F_ACC(slot), sub);
                        if (k == 0) WTL(12); // pelvis acceleration published
                    }
                    if (k == 0) { // integrate the root at once: the limbs that hang off it (the long chains) then have the next
                                  // substep's kinematics of their parent before the trunk's outward pass and integration are through
                        const v3 alpha = s6ang(acc_root), aO = s6lin(acc_root);
                        const v3 rv_new = rv + dt * (aO + cross(rw, rv));
                        v3 rw_new = rw + dt * alpha;
                        rw_new = DYN_RCP(1.f + dt * M.ang_damping) * rw_new;
                        const float wm = DYN_SQRT(dot(rw_new, rw_new));
                        if (wm > M.max_ang_vel) rw_new = (M.max_ang_vel * DYN_RCP(wm)) * rw_new;
                        rv = rv_new; rw = rw_new;
                        rp = rp + dt * rv;
                        rq = qnormalize(qmul(qexp(dt * rw), rq));
                        if (slot >= 0 && sub + 1 < nsub) {
                            s_rootp[0] = rp.x; s_rootp[64] = rp.y; s_rootp[128] = rp.z;
                            float *s = s_attkin + slot * 13 * 64;
                            s[0] = rq.x; s[64] = rq.y; s[128] = rq.z; s[192] = rq.w; s[256] = 0.f; s[320] = 0.f; s[384] = 0.f;
                            s[448] = rw.x; s[512] = rw.y; s[576] = rw.z; s[640] = rv.x; s[704] = rv.y; s[768] = rv.z;
                            publish(WV_F_KIN(slot), sub + 1);
                        }
                    }
                }
            }
            PARC_UNROLL
            for (int k = 0; k < WV_MAXLEN; ++k) if (k < tlen) wv_integrate_joint(M, W, W.body[0][k], trunk[k], dt);
            WSTAMP(7);
            WTL(13); // trunk outward + integration done
            if (sub + 1 < nsub) trunk_kinematics(sub + 1); // before this wave's own outward pass: the other limbs wait for it
            WTL(14); // next substep's trunk kinematics published
            WSTAMP(1);
        }
        // ---- outward pass + integration of this wave's limb ----------------------------------------------------------------
        if (has_limb) {
            if (w != 0) await(WV_F_ACC(W.par_slot[lc]), sub);
            WSTAMP(8);
            if (w != 0) WTL(6); // parent acceleration received
            s6 ap;
            PARC_UNROLL
            for (int a = 0; a < 6; ++a) ap.a[a] = s_attacc[(W.par_slot[lc] * 6 + a) * 64];
            PARC_UNROLL
            for (int k = 0; k < WV_MAXLEN; ++k) {
                if (k < llen) {
                    const int b = W.body[lc][k];
                    wv_joint_outward<1>(M, W, b, limb[k], ap, fac_l[k]);
                    wv_integrate_joint(M, W, b, limb[k], dt);
                }
            }
        }
        WSTAMP(9);
        WTL(15); // outward pass + integration of the own limb done
    }
    // ---- write back -----------------------------------------------------------------------------------------------------
    // Wave 0 is the last wave out of the substep loop.  The prep-record work on its state -- the quaternions of the trunk joints (two
    // spherical joints for the humanoid: ~800 instructions) and the heading terms of the root (atan2, two sin / cos pairs) -- is formed by
    // two other waves (WaveTables::ep_*), which are through their own stores by then: wave 0 hands the dofs and the root rotation over in
    // LDS (the attach-kinematics slots are dead by now) as its first action.
    const bool hand = prep != nullptr && W.helper >= 1;
    float *s_hand = s_attkin;
    if (hand && w == 0) {
        PARC_UNROLL
        for (int k = 0; k < WV_MAXLEN; ++k) {
            if (k < tlen && W.body[0][k] != 0) {
                const int jt = W.c[W.body[0][k]].jtype;
                const q4 jq = trunk[k].jq; const float hang = trunk[k].hang;
                if (jt == DJ_SPHERICAL) { const v3 ex = qlog(jq); s_hand[(3 * k) * 64] = ex.x; s_hand[(3 * k + 1) * 64] = ex.y; s_hand[(3 * k + 2) * 64] = ex.z; }
                else if (jt == DJ_HINGE) s_hand[(3 * k) * 64] = hang;
            }
        }
        s_hand[(3 * WV_MAXLEN) * 64] = rq.x; s_hand[(3 * WV_MAXLEN + 1) * 64] = rq.y; s_hand[(3 * WV_MAXLEN + 2) * 64] = rq.z; s_hand[(3 * WV_MAXLEN + 3) * 64] = rq.w;
        publish(WV_F_HAND, 0);
    }
    if (!env_ok) return;
    float *odp = buf.char_dof_pos + (size_t)D_ * e, *odv = buf.char_dof_vel + (size_t)D_ * e, *ocf = buf.contact_forces + 3 * (size_t)e * B_;
    float4 *pr = prep ? prep + (size_t)e * 16 : nullptr;
    // slot 0 of the prep record: heading terms of the new root rotation (k_env_prep, lane 0)
    auto heading_terms = [&](q4 q) __attribute__((always_inline)) {
        const float heading = parc::calc_heading(make_float4(q.x, q.y, q.z, q.w));
        const float4 hinv = parc::heading_quat_inv(heading);
        prep[(size_t)e * 16] = make_float4(cosf(heading), sinf(heading), hinv.z, hinv.w);
    };
    if (w == 0) {
        const v3 out = mk(anc.x + rp.x, anc.y + rp.y, anc.z + rp.z);
        float *o = buf.char_root_pos + 3 * (size_t)e; o[0] = out.x; o[1] = out.y; o[2] = out.z;
        if (root_shadow) {
            float *sh = root_shadow + 6 * (size_t)e;
            sh[0] = out.x; sh[1] = out.y; sh[2] = out.z;
            sh[3] = rp.x - (out.x - anc.x); sh[4] = rp.y - (out.y - anc.y); sh[5] = rp.z - (out.z - anc.z);
        }
        o = buf.char_root_rot + 4 * (size_t)e; o[0] = rq.x; o[1] = rq.y; o[2] = rq.z; o[3] = rq.w;
        if (prep && !hand) heading_terms(rq);
        o = buf.char_root_vel + 3 * (size_t)e; o[0] = rv.x; o[1] = rv.y; o[2] = rv.z;
        o = buf.char_root_ang_vel + 3 * (size_t)e; o[0] = rw.x; o[1] = rw.y; o[2] = rw.z;
    }
    PARC_UNROLL
    for (int k = 0; k < WV_MAXLEN; ++k) {
        if (k < llen) wv_store_joint(M, W, W.body[lc][k], limb[k], odp, odv, ocf, pr);
        if (k < tlen) wv_store_joint(M, W, W.body[0][k], trunk[k], odp, odv, ocf, hand ? nullptr : pr);
    }
    bool mine_ep = w == W.ep_head_wave;
    PARC_UNROLL
    for (int k = 1; k < WV_MAXLEN; ++k) mine_ep = mine_ep || (k < W.len[0] && W.ep_trunk_wave[k] == w);
    if (hand && mine_ep) {
        await(WV_F_HAND, 0);
        {
            PARC_UNROLL
            for (int k = 1; k < WV_MAXLEN; ++k) {
                const int b = W.body[0][k];
                if (k < W.len[0] && W.ep_trunk_wave[k] == w) {
                    const int jt = W.c[b].jtype;
                    if (jt == DJ_SPHERICAL) pr[b] = parc::exp_map_to_quat(parc::mk3(s_hand[(3 * k) * 64], s_hand[(3 * k + 1) * 64], s_hand[(3 * k + 2) * 64]));
                    else if (jt == DJ_HINGE) pr[b] = parc::axis_angle_to_quat(parc::mk3(W.c[b].axis[0], W.c[b].axis[1], W.c[b].axis[2]), s_hand[(3 * k) * 64]);
                    else pr[b] = make_float4(0.f, 0.f, 0.f, 1.f);
                }
            }
        }
        if (w == W.ep_head_wave) {
            q4 q; q.x = s_hand[(3 * WV_MAXLEN) * 64]; q.y = s_hand[(3 * WV_MAXLEN + 1) * 64]; q.z = s_hand[(3 * WV_MAXLEN + 2) * 64]; q.w = s_hand[(3 * WV_MAXLEN + 3) * 64];
            heading_terms(q);
        }
    }
    // A flag wait of this block hit its bound: some wave integrated a stale record.  Every wave checks as its LAST action (a wave that
    // timed out sees its own mark; wave 0, which writes the root, is through after every wave it waited for) and overwrites the root
    // position of the block's envs with NaN: the observation kernel then produces NaN observations and rewards for them.
    if (__hip_atomic_load(&s_flag[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
        float *o = buf.char_root_pos + 3 * (size_t)e;
        const float qn = __builtin_nanf("");
        o[0] = qn; o[1] = qn; o[2] = qn;
    }
#ifdef PARC_STAMPS
    WSTAMP(10);
    if (lane == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_wave_stamps[w][i], wacc[i]);
#endif
}
__global__ __launch_bounds__(256, 1) void k_dynamics_wave(const DynModel *__restrict__ Mp, const WaveTables *__restrict__ Wp, DynTerrain T,
                                                          ParcEnvBuffers buf, const float *__restrict__ action,
                                                          const float *__restrict__ env_off_all, float *__restrict__ root_shadow,
                                                          float4 *__restrict__ prep, float *__restrict__ man_g, int N, int epb) {
    dynamics_wave_body<false>(Mp, Wp, T, buf, action, env_off_all, root_shadow, prep, man_g, N, epb);
}
// control modes vel / torque / pd_exp / pd_1d (DynModel::ctrl)
__global__ __launch_bounds__(256, 1) void k_dynamics_wave_ff(const DynModel *__restrict__ Mp, const WaveTables *__restrict__ Wp, DynTerrain T,
                                                             ParcEnvBuffers buf, const float *__restrict__ action,
                                                             const float *__restrict__ env_off_all, float *__restrict__ root_shadow,
                                                             float4 *__restrict__ prep, float *__restrict__ man_g, int N, int epb) {
    dynamics_wave_body<true>(Mp, Wp, T, buf, action, env_off_all, root_shadow, prep, man_g, N, epb);
}
#endif // __HIPCC__

} // namespace parcdyn

#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
