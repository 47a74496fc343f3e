// parc_dynamics_coop.hpp — chain-parallel version of the dynamics step for gfx950 (HIP only).
//
// Same equations as parc_dynamics.hpp (which stays the readable reference, the host build for the tests and the
// thread-per-env fallback kernel), different mapping: 8 lanes cooperate on one env, each lane owning one kinematic
// CHAIN of <= 4 bodies (humanoid: trunk [pelvis, torso, head] + 2 arms + 2 legs), 8 envs per wavefront.
//   * the per-body state (orientation, origin, spatial velocity, joint-space factors) lives in LDS, the joint state
//     in registers; nothing goes to scratch (the thread-per-env kernel keeps ~8 KB per lane in scratch, which at
//     65 536 envs is a 540 MB working set streamed from HBM);
//   * chains exchange data through LDS at the attachment bodies only: parent kinematics down (13 floats), the
//     articulated inertia + bias of a finished limb up (27 floats), the parent's spatial acceleration down (6);
//   * a wave carries 8 envs instead of 64, so 65 536 envs are 8 192 waves (8 per SIMD) instead of 1 024.
// Chains run level by level (limbs, then trunk, on the way in; trunk, then limbs, on the way out), so a wave
// executes ~6 body-units of work per pass instead of 15.
#pragma once
#include "parc_dynamics.hpp"

#pragma clang fp contract(fast)

namespace parcdyn {

#define CO_MAXCH 8   // chains (= lanes) per env
#define CO_MAXLEN 4  // bodies per chain
#define CO_MAXUP 5   // chains hanging off other chains (slots for the limb -> parent hand-over)
#define CO_ENVS 8    // envs per 64-lane block

struct CoopTables {
    int nchain, nlevel;
    int len[CO_MAXCH];
    int body[CO_MAXCH][CO_MAXLEN];
    int level[CO_MAXCH];
    int par_body[CO_MAXCH];            // body the chain hangs off, -1 for the chain that starts at the root
    int npt[DYN_MAXB];                 // collision points per body (a contiguous range of the model's list)
    int pt0[DYN_MAXB];
    int nchild[DYN_MAXB];              // chains attached to a body (other than the continuation of its own chain)
    int child[DYN_MAXB][4];
    float brad[DYN_MAXB];              // bounding radius of the body's collision spheres about the body origin
    int nsg[DYN_MAXB];                 // collision segments per body (a contiguous range of the model's list; their ends are collision points)
    int sg0[DYN_MAXB];
};

// host: split the tree into chains of at most `cap` bodies, leaves first (humanoid: [0,1,2],[3,4,5],[6,7,8],[9,10,11],[12,13,14])
inline bool build_coop_tables(const DynModel &M, CoopTables &C) {
    memset(&C, 0, sizeof(C));
    const int B = M.B;
    int cap = (B + 4) / 5;
    if (cap < 2) cap = 2;
    if (cap > CO_MAXLEN) cap = CO_MAXLEN;
    int depth[DYN_MAXB], owner[DYN_MAXB], nkids[DYN_MAXB];
    for (int b = 0; b < B; ++b) { depth[b] = b == 0 ? 0 : depth[M.parent[b]] + 1; owner[b] = -1; nkids[b] = 0; }
    for (int b = 1; b < B; ++b) nkids[M.parent[b]]++;
    int nchain = 0;
    for (;;) { // deepest unassigned body whose children are all assigned = next chain tail
        int tail = -1;
        for (int b = 0; b < B; ++b) {
            if (owner[b] >= 0) continue;
            bool kids_done = true;
            for (int c = 1; c < B; ++c) if (M.parent[c] == b && owner[c] < 0) kids_done = false;
            if (kids_done && (tail < 0 || depth[b] > depth[tail])) tail = b;
        }
        if (tail < 0) break;
        if (nchain >= CO_MAXCH || nchain > CO_MAXUP) return false; // chain 0 needs no hand-over slot
        int tmp[CO_MAXLEN], n = 0, b = tail;
        while (b >= 0 && owner[b] < 0 && n < cap) {
            // do not walk up into a parent that still has other unassigned subtrees hanging off it unless it is ours to take last
            tmp[n++] = b; owner[b] = nchain;
            const int p = b == 0 ? -1 : M.parent[b];
            if (p < 0) break;
            bool siblings_pending = false;
            for (int c = 1; c < B; ++c) if (M.parent[c] == p && owner[c] < 0) siblings_pending = true;
            if (siblings_pending) break;
            b = p;
        }
        C.len[nchain] = n;
        for (int k = 0; k < n; ++k) C.body[nchain][k] = tmp[n - 1 - k];
        nchain++;
    }
    C.nchain = nchain;
    // the chain that contains the root must be lane 0 (it owns the root state)
    for (int c = 0; c < nchain; ++c) if (C.body[c][0] == 0 && c != 0) {
        int l = C.len[0]; C.len[0] = C.len[c]; C.len[c] = l;
        for (int k = 0; k < CO_MAXLEN; ++k) { int t = C.body[0][k]; C.body[0][k] = C.body[c][k]; C.body[c][k] = t; }
    }
    int chain_of[DYN_MAXB];
    for (int c = 0; c < nchain; ++c) for (int k = 0; k < C.len[c]; ++k) chain_of[C.body[c][k]] = c;
    if (C.body[0][0] != 0) return false;
    C.nlevel = 0;
    for (int c = 0; c < nchain; ++c) C.par_body[c] = C.body[c][0] == 0 ? -1 : M.parent[C.body[c][0]];
    for (int c = 0; c < nchain; ++c) { // level = number of chain hops to the root chain
        int l = 0, cc = c;
        while (C.par_body[cc] >= 0) { cc = chain_of[C.par_body[cc]]; ++l; if (l > CO_MAXCH) return false; }
        C.level[c] = l;
        if (l + 1 > C.nlevel) C.nlevel = l + 1;
    }
    for (int c = 0; c < nchain; ++c) if (C.par_body[c] >= 0) {
        const int pb = C.par_body[c];
        if (C.nchild[pb] >= 4) return false;
        C.child[pb][C.nchild[pb]++] = c;
    }
    for (int k = 0; k < M.ncol; ++k) {
        const int b = M.col_body[k];
        if (C.npt[b] == 0) C.pt0[b] = k;
        if (k != C.pt0[b] + C.npt[b]) return false; // points of a body must be contiguous
        C.npt[b]++;
        const float rr = sqrtf(M.col_pos[k][0] * M.col_pos[k][0] + M.col_pos[k][1] * M.col_pos[k][1] + M.col_pos[k][2] * M.col_pos[k][2]);
        if (rr + M.col_r[k] > C.brad[b]) C.brad[b] = rr + M.col_r[k];
    }
    for (int k = 0; k < M.nseg; ++k) {
        const int b = M.seg_body[k];
        if (C.nsg[b] == 0) C.sg0[b] = k;
        if (k != C.sg0[b] + C.nsg[b]) return false; // segments of a body must be contiguous
        C.nsg[b]++;
    }
    return true;
}

#if defined(__HIPCC__)

// per-body joint-space factors kept in LDS between the inward and the outward pass:
// [0,6) cJ  [6,24) U (3 columns)  [24,30) Dinv  [30,33) u
#define CO_JNT 33
#define CO_MAXM 48  // contact planes of one chain (4 per candidate at most; the humanoid's legs have 16 candidates)

#ifdef PARC_STAMPS
__device__ unsigned long long g_dyn_stamps[16];
#define DSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); dacc[i] += t_ - dlast; dlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(64, 1) void k_dynamics_coop(const DynModel *__restrict__ Mp, const CoopTables *__restrict__ Cp, DynTerrain T,
                                                         ParcEnvBuffers buf, const float *__restrict__ action,
                                                         const float *__restrict__ env_off_all, int N) {
    // model tables are indexed per lane (body / dof / collision point differ between the lanes of a wave), so they are
    // staged in LDS once per block: a global load per table access would stall the single resident wave of a SIMD
    __shared__ int s_model[(sizeof(DynModel) + 3) / 4], s_coop[(sizeof(CoopTables) + 3) / 4];
    for (int i = threadIdx.x; i < (int)(sizeof(DynModel) / 4); i += 64) s_model[i] = reinterpret_cast<const int *>(Mp)[i];
    for (int i = threadIdx.x; i < (int)(sizeof(CoopTables) / 4); i += 64) s_coop[i] = reinterpret_cast<const int *>(Cp)[i];
    __syncthreads();
    const DynModel &M = *reinterpret_cast<const DynModel *>(s_model);
    const CoopTables &C = *reinterpret_cast<const CoopTables *>(s_coop);
    const DynModel &Mg = *Mp;   // wave-uniform scalars stay scalar loads
    const CoopTables &Cg = *Cp;
    __shared__ float s_kin[CO_ENVS][DYN_MAXB * 14 + 1];  // bq4 r3 vel6
    __shared__ float s_up[CO_ENVS][CO_MAXUP * 28 + 1];   // Ia(21) + pa(6) of a finished chain (slot = chain - 1)
    __shared__ float s_jnt[CO_ENVS][DYN_MAXB * CO_JNT + 1];
    __shared__ float s_patch[CO_ENVS][DYN_PATCH * DYN_PATCH];
    __shared__ float s_pmax[CO_ENVS][(DYN_PATCH - 4) * (DYN_PATCH - 4)]; // max height over the 5x5 cells around a cell (+inf near the patch border)
    __shared__ int s_pox[CO_ENVS], s_poy[CO_ENVS];
    __shared__ float s_rootp[CO_ENVS][4];

    const int lane = threadIdx.x, c = lane & 7, el = lane >> 3;
#ifdef PARC_STAMPS
    unsigned long long dacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dlast = __builtin_readcyclecounter();
#endif
    const int e = blockIdx.x * CO_ENVS + el;
    const bool env_ok = e < N;
    const bool active = env_ok && c < Cg.nchain;
    const int ec = env_ok ? e : N - 1; // clamp for address safety; results of inactive lanes are never stored
    const float dt = Mg.dt;
    const int B = Mg.B;
    const int my_len = active ? C.len[c] : 0;
    const int my_level = active ? C.level[c] : -1;
    const int my_par = active ? C.par_body[c] : -1;
    const float eo0 = env_off_all[3 * ec], eo1 = env_off_all[3 * ec + 1], eo2 = env_off_all[3 * ec + 2];

    // ---- persistent joint state of this lane's bodies ---------------------------------------------------------
    q4 jq[CO_MAXLEN], tq[CO_MAXLEN];
    float hang[CO_MAXLEN], thang[CO_MAXLEN];
    v3 qd[CO_MAXLEN];
    int bid[CO_MAXLEN], jty[CO_MAXLEN], did[CO_MAXLEN];
#pragma unroll
    for (int k = 0; k < CO_MAXLEN; ++k) {
        const int b = k < my_len ? C.body[c][k] : 0;
        bid[k] = b; jty[k] = k < my_len ? M.jtype[b] : DJ_FIXED; did[k] = M.dof_idx[b];
        jq[k].x = 0.f; jq[k].y = 0.f; jq[k].z = 0.f; jq[k].w = 1.f; tq[k] = jq[k]; hang[k] = 0.f; thang[k] = 0.f; qd[k] = mk(0.f, 0.f, 0.f);
        const float *dp = buf.char_dof_pos + (size_t)Mg.D * ec, *dv = buf.char_dof_vel + (size_t)Mg.D * ec, *ac = action + (size_t)Mg.D * ec;
        const int di = did[k];
        if (jty[k] == DJ_SPHERICAL) {
            jq[k] = qexp(mk(dp[di], dp[di + 1], dp[di + 2]));
            qd[k] = mk(dv[di], dv[di + 1], dv[di + 2]);
            if (Mg.ctrl == PARC_CTRL_PD) {
                tq[k] = qexp(mk(clampf(ac[di], M.act_lo[di], M.act_hi[di]), clampf(ac[di + 1], M.act_lo[di + 1], M.act_hi[di + 1]),
                                clampf(ac[di + 2], M.act_lo[di + 2], M.act_hi[di + 2])));
            } else { // the other control modes keep the joint's feed-forward torque (ctrl_ff) where the pd mode keeps its target
                v3 diff = mk(0.f, 0.f, 0.f);
                if (Mg.ctrl >= PARC_CTRL_PD_EXP) diff = qlog(qmul(qconj(jq[k]), qexp(mk(ac[di], ac[di + 1], ac[di + 2]))));
                tq[k].x = ctrl_ff(Mg.ctrl, M.kp[di], M.kd[di], M.eff[di], M.act_lo[di], M.act_hi[di], ac[di], diff.x, qd[k].x);
                tq[k].y = ctrl_ff(Mg.ctrl, M.kp[di + 1], M.kd[di + 1], M.eff[di + 1], M.act_lo[di + 1], M.act_hi[di + 1], ac[di + 1], diff.y, qd[k].y);
                tq[k].z = ctrl_ff(Mg.ctrl, M.kp[di + 2], M.kd[di + 2], M.eff[di + 2], M.act_lo[di + 2], M.act_hi[di + 2], ac[di + 2], diff.z, qd[k].z);
            }
        } else if (jty[k] == DJ_HINGE) {
            hang[k] = dp[di]; qd[k].x = dv[di];
            if (Mg.ctrl == PARC_CTRL_PD) thang[k] = clampf(ac[di], M.act_lo[di], M.act_hi[di]);
            else thang[k] = ctrl_ff(Mg.ctrl, M.kp[di], M.kd[di], M.eff[di], M.act_lo[di], M.act_hi[di], ac[di], ctrl_hinge_diff(Mg.ctrl, ac[di], hang[k]), qd[k].x);
        }
    }
    // root state lives in the lane of chain 0
    v3 rp = mk(buf.char_root_pos[3 * ec], buf.char_root_pos[3 * ec + 1], buf.char_root_pos[3 * ec + 2]);
    q4 rq; rq.x = buf.char_root_rot[4 * ec]; rq.y = buf.char_root_rot[4 * ec + 1]; rq.z = buf.char_root_rot[4 * ec + 2]; rq.w = buf.char_root_rot[4 * ec + 3];
    rq = qnormalize(rq);
    v3 rv = mk(buf.char_root_vel[3 * ec], buf.char_root_vel[3 * ec + 1], buf.char_root_vel[3 * ec + 2]);
    v3 rw = mk(buf.char_root_ang_vel[3 * ec], buf.char_root_ang_vel[3 * ec + 1], buf.char_root_ang_vel[3 * ec + 2]);

    // ---- local height patch, loaded by the 8 lanes of the env ------------------------------------------------------
    {
        const int pox = cell_of(rp.x + eo0, T.min_x, T.dx) - DYN_PATCH / 2, poy = cell_of(rp.y + eo1, T.min_y, T.dy) - DYN_PATCH / 2;
        if (c == 0) { s_pox[el] = pox; s_poy[el] = poy; }
        for (int i = c; i < DYN_PATCH * DYN_PATCH; i += 8) s_patch[el][i] = hf_at(T, pox + i / DYN_PATCH, poy + i % DYN_PATCH);
    }
    v3 fcon[CO_MAXLEN];
#pragma unroll
    for (int k = 0; k < CO_MAXLEN; ++k) fcon[k] = mk(0.f, 0.f, 0.f);
    // contact planes of this lane's chain (parc_dynamics.hpp, contact manifold): private memory -- this kernel is the general fallback and
    // the cross-check of k_dynamics_wave, not the fast path
    ContactPlane man[CO_MAXM];
    int man_n = 0, man_start[CO_MAXLEN], man_end[CO_MAXLEN];
    for (int k = 0; k < CO_MAXLEN; ++k) { man_start[k] = 0; man_end[k] = 0; }
    v3 rootp_d = mk(0.f, 0.f, 0.f);
    __syncthreads();
    // a body whose bounding sphere clears every column its collision spheres could touch (own cell +-1 for the sphere
    // centres, +-1 more for the neighbour columns) skips the contact loop; exact, since those contributions are zero
    for (int i = c; i < (DYN_PATCH - 4) * (DYN_PATCH - 4); i += 8) {
        const int pi_ = i / (DYN_PATCH - 4) + 2, pj_ = i % (DYN_PATCH - 4) + 2;
        float m = -3.0e38f;
        for (int a = -2; a <= 2; ++a) for (int q = -2; q <= 2; ++q) m = fmaxf(m, s_patch[el][(pi_ + a) * DYN_PATCH + pj_ + q]);
        s_pmax[el][i] = m;
    }
    const float cell_min = fminf(T.dx, T.dy);
    __syncthreads();

    DSTAMP(0);
    for (int sub = 0; sub < Mg.nsub; ++sub) {
        if (c == 0) { s_rootp[el][0] = rp.x; s_rootp[el][1] = rp.y; s_rootp[el][2] = rp.z; }
        // ================= kinematics, root chain first =================
        for (int lv = 0; lv < Cg.nlevel; ++lv) {
            if (my_level == lv) {
                q4 pq = rq; v3 pr = mk(0.f, 0.f, 0.f); s6 pv = s6mk(rw, rv);
                if (my_par >= 0) {
                    const float *s = &s_kin[el][my_par * 14];
                    pq.x = s[0]; pq.y = s[1]; pq.z = s[2]; pq.w = s[3]; pr = mk(s[4], s[5], s[6]);
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) pv.a[a] = s[7 + a];
                }
#pragma unroll
                for (int k = 0; k < CO_MAXLEN; ++k) {
                    if (k < my_len) {
                        const int b = bid[k];
                        s6 cJ = s6zero();
                        if (b != 0) {
                            const m3 Rp = qmat(pq);
                            pr = pr + mulv(Rp, mk(M.lt[b][0], M.lt[b][1], M.lt[b][2]));
                            q4 lq; lq.x = M.lr[b][0]; lq.y = M.lr[b][1]; lq.z = M.lr[b][2]; lq.w = M.lr[b][3];
                            q4 jr = jq[k];
                            if (jty[k] == DJ_HINGE) jr = qexp(hang[k] * mk(M.axis[b][0], M.axis[b][1], M.axis[b][2]));
                            pq = qnormalize(qmul(pq, qmul(lq, jr)));
                            const m3 R = qmat(pq);
                            v3 wj = mk(0.f, 0.f, 0.f);
                            if (jty[k] == DJ_SPHERICAL) wj = mulv(R, qd[k]);
                            else if (jty[k] == DJ_HINGE) wj = qd[k].x * mulv(R, mk(M.axis[b][0], M.axis[b][1], M.axis[b][2]));
                            const s6 vJ = s6mk(wj, cross(pr, wj));
                            pv = pv + vJ;
                            cJ = crm(pv, vJ);
                        }
                        float *s = &s_kin[el][b * 14];
                        s[0] = pq.x; s[1] = pq.y; s[2] = pq.z; s[3] = pq.w; s[4] = pr.x; s[5] = pr.y; s[6] = pr.z;
                        PARC_UNROLL
                        for (int a = 0; a < 6; ++a) { s[7 + a] = pv.a[a]; s_jnt[el][b * CO_JNT + a] = cJ.a[a]; }
                    }
                }
            }
            __syncthreads();
        }
        DSTAMP(1);
        // ================= inward pass, outermost chains first =================
        s6 acc_root = s6zero();
        for (int lv = Cg.nlevel - 1; lv >= 0; --lv) {
            if (my_level == lv) {
                sym6 Ic; s6 pc; // carry from the child body of this chain
                PARC_UNROLL
                for (int i = 0; i < 21; ++i) Ic.s[i] = 0.f;
                pc = s6zero();
                const v3 rootp = mk(s_rootp[el][0], s_rootp[el][1], s_rootp[el][2]);
#pragma unroll 1
                for (int k = my_len - 1; k >= 0; --k) {
                    {
                        // joint state of position k (registers; selected, not indexed, so nothing goes to scratch)
                        int b = bid[0], di = did[0], jt = jty[0];
                        q4 jqk = jq[0], tqk = tq[0]; float hk = hang[0], thk = thang[0]; v3 qdk = qd[0];
#pragma unroll
                        for (int q = 1; q < CO_MAXLEN; ++q)
                            if (k == q) { b = bid[q]; di = did[q]; jt = jty[q]; jqk = jq[q]; tqk = tq[q]; hk = hang[q]; thk = thang[q]; qdk = qd[q]; }
                        const float *sk = &s_kin[el][b * 14];
                        q4 bqk; bqk.x = sk[0]; bqk.y = sk[1]; bqk.z = sk[2]; bqk.w = sk[3];
                        const m3 R = qmat(bqk);
                        const v3 r = mk(sk[4], sk[5], sk[6]);
                        s6 velk, cJk;
                        PARC_UNROLL
                        for (int a = 0; a < 6; ++a) { velk.a[a] = sk[7 + a]; cJk.a[a] = s_jnt[el][b * CO_JNT + a]; }
                        float *sj = &s_jnt[el][b * CO_JNT];
                        sym6 IA = Ic;
                        s6 pA = pc;
                        DSTAMP(2);
                        // own spatial inertia about O, velocity-product bias, gravity
                        {
                            const v3 cm = r + mulv(R, mk(M.com[b][0], M.com[b][1], M.com[b][2]));
                            const float Ib[3][3] = {{M.inertia[b][0], M.inertia[b][3], M.inertia[b][4]}, {M.inertia[b][3], M.inertia[b][1], M.inertia[b][5]},
                                                    {M.inertia[b][4], M.inertia[b][5], M.inertia[b][2]}};
                            float RI[3][3], Iw[3][3];
                            PARC_UNROLL
                            for (int a = 0; a < 3; ++a) PARC_UNROLL for (int q = 0; q < 3; ++q) RI[a][q] = R.m[a][0] * Ib[0][q] + R.m[a][1] * Ib[1][q] + R.m[a][2] * Ib[2][q];
                            PARC_UNROLL
                            for (int a = 0; a < 3; ++a) PARC_UNROLL for (int q = 0; q < 3; ++q) Iw[a][q] = RI[a][0] * R.m[q][0] + RI[a][1] * R.m[q][1] + RI[a][2] * R.m[q][2];
                            const float Icw[6] = {Iw[0][0], Iw[1][1], Iw[2][2], Iw[0][1], Iw[0][2], Iw[1][2]};
                            sym6 Iown;
                            PARC_UNROLL
                            for (int i = 0; i < 21; ++i) Iown.s[i] = 0.f;
                            add_inertia(Iown, M.mass[b], cm, Icw);
                            const s6 Iv = symmul(Iown, velk);
                            const s6 pb = crf(velk, Iv);
                            const v3 fg = mk(0.f, 0.f, M.mass[b] * Mg.gravity_z);
                            const v3 ng = cross(cm, fg);
                            PARC_UNROLL
                            for (int i = 0; i < 21; ++i) IA.s[i] += Iown.s[i];
                            pA.a[0] += pb.a[0] - ng.x; pA.a[1] += pb.a[1] - ng.y; pA.a[2] += pb.a[2] - ng.z;
                            pA.a[3] += pb.a[3] - fg.x; pA.a[4] += pb.a[4] - fg.y; pA.a[5] += pb.a[5] - fg.z;
                        }
                        DSTAMP(3);
                        // contacts of this body: discovery into this lane's planes (substeps 0, man_period, ...; parc_dynamics.hpp, contact
                        // manifold), then the evaluation of the body's planes
                        v3 fsum = mk(0.f, 0.f, 0.f);
                        if (sub % Mg.man_period == 0) {
                            if (k == my_len - 1) { man_n = 0; rootp_d = rootp; }
                            man_start[k] = man_n;
                            int npt_b = C.npt[b];
                            {   // exact cull: the body's bounding sphere (+ the largest margin) clears every column its spheres could reach
                                const int bx = cell_of(r.x + rootp.x + eo0, T.min_x, T.dx) - s_pox[el], by = cell_of(r.y + rootp.y + eo1, T.min_y, T.dy) - s_poy[el];
                                if (C.brad[b] < cell_min && bx >= 2 && bx < DYN_PATCH - 2 && by >= 2 && by < DYN_PATCH - 2 &&
                                    r.z + rootp.z + eo2 - C.brad[b] - Mg.spec_max > s_pmax[el][(bx - 2) * (DYN_PATCH - 4) + by - 2]) npt_b = 0;
                            }
                            auto top_at = [&](int ix, int iy) {
                                const int a_ = ix - s_pox[el], b_ = iy - s_poy[el];
                                return (a_ >= 0 && a_ < DYN_PATCH && b_ >= 0 && b_ < DYN_PATCH) ? s_patch[el][a_ * DYN_PATCH + b_] : hf_at(T, ix, iy); };
                            const v3 goff = mk(rootp.x + eo0, rootp.y + eo1, rootp.z + eo2);
                            auto candidate = [&](v3 x, float rad, float w) {
                                const v3 vpt = s6lin(velk) + cross(s6ang(velk), x);
                                sphere_discover(Mg, T, x + goff, rad, vpt, top_at, [&](float pen, v3 n) {
                                    if (man_n >= CO_MAXM) return;
                                    ContactPlane &P = man[man_n++];
                                    P.body = k; P.p = mulTv(R, x - r); P.n = n; P.off = pen + dot(n, x); P.w = w;
                                });
                            };
                            for (int pi = 0; pi < npt_b; ++pi) {
                                const int kp = C.pt0[b] + pi;
                                candidate(r + mulv(R, mk(M.col_pos[kp][0], M.col_pos[kp][1], M.col_pos[kp][2])), M.col_r[kp], 1.f);
                            }
                            // shafts of capsules / sole edges against the columns' top edges (segment_edge_point); the segments' ends are
                            // collision points, so the body-level cull above covers them
                            const int nsg_b = npt_b > 0 ? C.nsg[b] : 0;
                            for (int si = 0; si < nsg_b; ++si) {
                                const int ks = C.sg0[b] + si;
                                const v3 xa = r + mulv(R, mk(M.seg_a[ks][0], M.seg_a[ks][1], M.seg_a[ks][2]));
                                const v3 xb = r + mulv(R, mk(M.seg_b[ks][0], M.seg_b[ks][1], M.seg_b[ks][2]));
                                v3 Q;
                                const float wq = segment_edge_point(T, xa + goff, xb + goff, top_at, Q);
                                if (wq > 0.f) candidate(Q - goff, M.seg_r[ks], wq);
                            }
                            man_end[k] = man_n;
                        }
                        {
                            const v3 shift = rootp - rootp_d;
                            for (int ci = man_start[k]; ci < man_end[k]; ++ci) {
                                const ContactPlane &P = man[ci];
                                const v3 x = r + mulv(R, P.p);
                                const float pen = P.off - dot(P.n, x + shift);
                                if (!(pen > 0.f)) continue;
                                const v3 vpt = s6lin(velk) + cross(s6ang(velk), x);
                                contact_apply(Mg, dt, x, vpt, pen, P.n, IA, pA, fsum, P.w);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < CO_MAXLEN; ++q) if (k == q) fcon[q] = fsum;
                        DSTAMP(4);
                        // finished limbs hanging off this body
                        for (int ci = 0; ci < C.nchild[b]; ++ci) {
                            const float *s = &s_up[el][(C.child[b][ci] - 1) * 28];
                            PARC_UNROLL
                            for (int i = 0; i < 21; ++i) IA.s[i] += s[i];
                            PARC_UNROLL
                            for (int a = 0; a < 6; ++a) pA.a[a] += s[21 + a];
                        }
                        DSTAMP(5);
                        if (b == 0) { // floating base: solve IA a0 = -pA (Cholesky)
                            float Lm[6][6];
                            PARC_UNROLL
                            for (int a = 0; a < 6; ++a) PARC_UNROLL for (int q = 0; q < 6; ++q) Lm[a][q] = 0.f;
                            PARC_UNROLL
                            for (int j = 0; j < 6; ++j) {
                                float sd = sget(IA, j, j);
                                PARC_UNROLL
                                for (int q = 0; q < j; ++q) sd -= Lm[j][q] * Lm[j][q];
                                sd = sd > 1e-12f ? sqrtf(sd) : 1e3f; // a non-positive pivot is a numerical breakdown: treat the direction as immovable (no acceleration) rather than as massless
                                Lm[j][j] = sd;
                                PARC_UNROLL
                                for (int a = j + 1; a < 6; ++a) {
                                    float sa = sget(IA, a, j);
                                    PARC_UNROLL
                                    for (int q = 0; q < j; ++q) sa -= Lm[a][q] * Lm[j][q];
                                    Lm[a][j] = sa / sd;
                                }
                            }
                            float y[6], xs[6];
                            PARC_UNROLL
                            for (int a = 0; a < 6; ++a) { float sa = -pA.a[a]; for (int q = 0; q < a; ++q) sa -= Lm[a][q] * y[q]; y[a] = sa / Lm[a][a]; }
                            PARC_UNROLL
                            for (int a = 5; a >= 0; --a) { float sa = y[a]; for (int q = a + 1; q < 6; ++q) sa -= Lm[q][a] * xs[q]; xs[a] = sa / Lm[a][a]; }
                            PARC_UNROLL
                            for (int a = 0; a < 6; ++a) acc_root.a[a] = xs[a];
                            DSTAMP(6);
                        } else {
                            const int nd = jt == DJ_SPHERICAL ? 3 : (jt == DJ_HINGE ? 1 : 0);
                            if (nd == 0) {
                                const s6 Iac = symmul(IA, cJk);
                                Ic = IA;
                                pc = pA + Iac;
                            } else {
                                s6 Uc[3]; float Dinvk[6], uuk[3];
                                s6 Sc[3];
                                float tau[3] = {0.f, 0.f, 0.f}, aug[3] = {0.f, 0.f, 0.f};
                                if (nd == 3) {
                                    PARC_UNROLL
                                    for (int q = 0; q < 3; ++q) { const v3 a = mk(R.m[0][q], R.m[1][q], R.m[2][q]); Sc[q] = s6mk(a, cross(r, a)); }
                                    const bool pd = Mg.ctrl == PARC_CTRL_PD;
                                    const v3 err = pd ? qlog(qmul(qconj(jqk), tqk)) : mk(tqk.x, tqk.y, tqk.z); // (not pd: the feed-forward torque)
                                    const v3 cur = qlog(jqk);
                                    const float e3[3] = {err.x, err.y, err.z}, c3[3] = {cur.x, cur.y, cur.z}, q3[3] = {qdk.x, qdk.y, qdk.z};
                                    PARC_UNROLL
                                    for (int q = 0; q < 3; ++q) {
                                        float t = M.kp[di + q] * e3[q] - (M.kd[di + q] + dt * M.kp[di + q]) * q3[q];
                                        t = clampf(t, -M.eff[di + q], M.eff[di + q]);
                                        aug[q] = M.arm[di + q] + dt * M.kd[di + q] + dt * dt * M.kp[di + q];
                                        if (!pd) drive_ff(Mg.ctrl, M.kd[di + q], M.arm[di + q], M.eff[di + q], dt, e3[q], q3[q], t, aug[q]);
                                        if (c3[q] < M.lo[di + q]) { t += Mg.lim_k * (M.lo[di + q] - c3[q]) - Mg.lim_d * q3[q]; aug[q] += dt * Mg.lim_d + dt * dt * Mg.lim_k; }
                                        else if (c3[q] > M.hi[di + q]) { t += Mg.lim_k * (M.hi[di + q] - c3[q]) - Mg.lim_d * q3[q]; aug[q] += dt * Mg.lim_d + dt * dt * Mg.lim_k; }
                                        tau[q] = t;
                                    }
                                } else {
                                    const v3 a = mulv(R, mk(M.axis[b][0], M.axis[b][1], M.axis[b][2]));
                                    Sc[0] = s6mk(a, cross(r, a)); Sc[1] = s6zero(); Sc[2] = s6zero();
                                    float t = M.kp[di] * (thk - hk) - (M.kd[di] + dt * M.kp[di]) * qdk.x;
                                    t = clampf(t, -M.eff[di], M.eff[di]);
                                    aug[0] = M.arm[di] + dt * M.kd[di] + dt * dt * M.kp[di];
                                    if (Mg.ctrl != PARC_CTRL_PD) drive_ff(Mg.ctrl, M.kd[di], M.arm[di], M.eff[di], dt, thk, qdk.x, t, aug[0]);
                                    if (hk < M.lo[di]) { t += Mg.lim_k * (M.lo[di] - hk) - Mg.lim_d * qdk.x; aug[0] += dt * Mg.lim_d + dt * dt * Mg.lim_k; }
                                    else if (hk > M.hi[di]) { t += Mg.lim_k * (M.hi[di] - hk) - Mg.lim_d * qdk.x; aug[0] += dt * Mg.lim_d + dt * dt * Mg.lim_k; }
                                    tau[0] = t;
                                }
                                float Dm[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
                                PARC_UNROLL
                                for (int q = 0; q < 3; ++q) {
                                    Uc[q] = s6zero(); uuk[q] = 0.f;
                                    if (q < nd) {
                                        Uc[q] = symmul(IA, Sc[q]);
                                        float sp = 0.f;
                                        PARC_UNROLL
                                        for (int a = 0; a < 6; ++a) sp += Sc[q].a[a] * pA.a[a];
                                        uuk[q] = tau[q] - sp;
                                    }
                                }
                                PARC_UNROLL
                                for (int q = 0; q < 3; ++q)
                                    PARC_UNROLL
                                    for (int l = 0; l < 3; ++l) {
                                        float a_ = 0.f;
                                        PARC_UNROLL
                                        for (int a = 0; a < 6; ++a) a_ += Sc[q].a[a] * Uc[l].a[a];
                                        if (q < nd && l < nd) Dm[q][l] = a_ + (q == l ? aug[q] : 0.f); // unused rows stay identity
                                    }
                                float *Di = Dinvk;
                                PARC_UNROLL
                                for (int q = 0; q < 6; ++q) Di[q] = 0.f;
                                if (nd == 1) Di[0] = 1.f / Dm[0][0];
                                else {
                                    const float c00 = Dm[1][1] * Dm[2][2] - Dm[1][2] * Dm[2][1], c01 = Dm[0][2] * Dm[2][1] - Dm[0][1] * Dm[2][2],
                                                c02 = Dm[0][1] * Dm[1][2] - Dm[0][2] * Dm[1][1];
                                    const float id = 1.f / (Dm[0][0] * c00 + Dm[1][0] * c01 + Dm[2][0] * c02);
                                    Di[0] = c00 * id; Di[3] = c01 * id; Di[4] = c02 * id;
                                    Di[1] = (Dm[0][0] * Dm[2][2] - Dm[0][2] * Dm[2][0]) * id;
                                    Di[5] = (Dm[0][2] * Dm[1][0] - Dm[0][0] * Dm[1][2]) * id;
                                    Di[2] = (Dm[0][0] * Dm[1][1] - Dm[0][1] * Dm[1][0]) * id;
                                }
                                s6 Kc[3];
                                float Ku[6];
                                if (nd == 1) {
                                    PARC_UNROLL
                                    for (int a = 0; a < 6; ++a) { Kc[0].a[a] = Uc[0].a[a] * Di[0]; Kc[1].a[a] = 0.f; Kc[2].a[a] = 0.f; Ku[a] = Kc[0].a[a] * uuk[0]; }
                                } else {
                                    const float D3[3][3] = {{Di[0], Di[3], Di[4]}, {Di[3], Di[1], Di[5]}, {Di[4], Di[5], Di[2]}};
                                    PARC_UNROLL
                                    for (int q = 0; q < 3; ++q)
                                        PARC_UNROLL
                                        for (int a = 0; a < 6; ++a) Kc[q].a[a] = Uc[0].a[a] * D3[0][q] + Uc[1].a[a] * D3[1][q] + Uc[2].a[a] * D3[2][q];
                                    PARC_UNROLL
                                    for (int a = 0; a < 6; ++a) Ku[a] = Kc[0].a[a] * uuk[0] + Kc[1].a[a] * uuk[1] + Kc[2].a[a] * uuk[2];
                                }
                                Ic = IA;
                                PARC_UNROLL
                                for (int a = 0; a < 6; ++a)
                                    PARC_UNROLL
                                    for (int q = a; q < 6; ++q) {
                                        float a_ = 0.f;
                                        PARC_UNROLL
                                        for (int l = 0; l < 3; ++l) a_ += Kc[l].a[a] * Uc[l].a[q];
                                        Ic.s[sidx(a, q)] -= a_;
                                    }
                                const s6 Iac = symmul(Ic, cJk);
                                PARC_UNROLL
                                for (int a = 0; a < 6; ++a) pc.a[a] = pA.a[a] + Iac.a[a] + Ku[a];
                                PARC_UNROLL
                                for (int q = 0; q < 3; ++q) { for (int a = 0; a < 6; ++a) sj[6 + 6 * q + a] = Uc[q].a[a]; sj[30 + q] = uuk[q]; }
                                PARC_UNROLL
                                for (int q = 0; q < 6; ++q) sj[24 + q] = Dinvk[q];
                            }
                            DSTAMP(7);
                        }
                    }
                }
                if (my_par >= 0) { // hand the chain's articulated inertia / bias to the parent chain
                    float *s = &s_up[el][(c - 1) * 28];
                    PARC_UNROLL
                    for (int i = 0; i < 21; ++i) s[i] = Ic.s[i];
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) s[21 + a] = pc.a[a];
                }
            }
            __syncthreads();
        }
        DSTAMP(8);
        // ================= outward pass: accelerations, root chain first =================
        v3 qdd[CO_MAXLEN];
#pragma unroll
        for (int k = 0; k < CO_MAXLEN; ++k) qdd[k] = mk(0.f, 0.f, 0.f);
        for (int lv = 0; lv < Cg.nlevel; ++lv) {
            if (my_level == lv) {
                s6 ap = acc_root;
                if (my_par >= 0) for (int a = 0; a < 6; ++a) ap.a[a] = s_kin[el][my_par * 14 + 7 + a];
#pragma unroll 1
                for (int k = 0; k < my_len; ++k) {
                    int b = bid[0], jt = jty[0];
#pragma unroll
                    for (int q = 1; q < CO_MAXLEN; ++q) if (k == q) { b = bid[q]; jt = jty[q]; }
                    s6 ai = ap;
                    if (b != 0) {
                        const float *sj = &s_jnt[el][b * CO_JNT];
                        const int nd = jt == DJ_SPHERICAL ? 3 : (jt == DJ_HINGE ? 1 : 0);
                        PARC_UNROLL
                        for (int a = 0; a < 6; ++a) ai.a[a] = ap.a[a] + sj[a];
                        if (nd > 0) {
                            float rhs[3];
                            PARC_UNROLL
                            for (int q = 0; q < 3; ++q) { // U / u of unused dofs are zero
                                float ua = 0.f;
                                PARC_UNROLL
                                for (int a = 0; a < 6; ++a) ua += sj[6 + 6 * q + a] * ai.a[a];
                                rhs[q] = sj[30 + q] - ua;
                            }
                            float q3[3] = {sj[24] * rhs[0], 0.f, 0.f};
                            if (nd == 3) {
                                q3[0] = sj[24] * rhs[0] + sj[27] * rhs[1] + sj[28] * rhs[2];
                                q3[1] = sj[27] * rhs[0] + sj[25] * rhs[1] + sj[29] * rhs[2];
                                q3[2] = sj[28] * rhs[0] + sj[29] * rhs[1] + sj[26] * rhs[2];
                            }
                            const v3 qv = mk(q3[0], q3[1], q3[2]);
#pragma unroll
                            for (int q = 0; q < CO_MAXLEN; ++q) if (k == q) qdd[q] = qv;
                            const float *sk = &s_kin[el][b * 14];
                            q4 bqk; bqk.x = sk[0]; bqk.y = sk[1]; bqk.z = sk[2]; bqk.w = sk[3];
                            const m3 R = qmat(bqk);
                            v3 wj;
                            if (nd == 3) wj = mulv(R, qv);
                            else wj = q3[0] * mulv(R, mk(M.axis[b][0], M.axis[b][1], M.axis[b][2]));
                            ai = ai + s6mk(wj, cross(mk(sk[4], sk[5], sk[6]), wj));
                        }
                    }
                    PARC_UNROLL
                    for (int a = 0; a < 6; ++a) s_kin[el][b * 14 + 7 + a] = ai.a[a];
                    ap = ai;
                }
            }
            __syncthreads();
        }
        DSTAMP(9);
        // ================= integrate =================
        if (c == 0) {
            const v3 alpha = s6ang(acc_root), aO = s6lin(acc_root);
            const v3 rv_new = rv + dt * (aO + cross(rw, rv));
            v3 rw_new = rw + dt * alpha;
            rw_new = (1.f / (1.f + dt * Mg.ang_damping)) * rw_new;
            const float wm = sqrtf(dot(rw_new, rw_new));
            if (wm > Mg.max_ang_vel) rw_new = (Mg.max_ang_vel / wm) * rw_new;
            rv = rv_new; rw = rw_new;
            rp = rp + dt * rv;
            rq = qnormalize(qmul(qexp(dt * rw), rq));
        }
#pragma unroll
        for (int k = 0; k < CO_MAXLEN; ++k) {
            if (k < my_len) {
                if (jty[k] == DJ_SPHERICAL) {
                    qd[k] = mk(clampf(qd[k].x + dt * qdd[k].x, -Mg.max_ang_vel, Mg.max_ang_vel), clampf(qd[k].y + dt * qdd[k].y, -Mg.max_ang_vel, Mg.max_ang_vel),
                               clampf(qd[k].z + dt * qdd[k].z, -Mg.max_ang_vel, Mg.max_ang_vel));
                    jq[k] = qnormalize(qmul(jq[k], qexp(dt * qd[k])));
                } else if (jty[k] == DJ_HINGE) {
                    qd[k].x = clampf(qd[k].x + dt * qdd[k].x, -Mg.max_ang_vel, Mg.max_ang_vel);
                    hang[k] += dt * qd[k].x;
                }
            }
        }
        __syncthreads();
        DSTAMP(10);
    }
    // ---- write back -------------------------------------------------------------------------------------------------
#ifdef PARC_STAMPS
    if (lane == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_dyn_stamps[i], dacc[i]);
#endif
    if (!active) return;
    if (c == 0) {
        float *o = buf.char_root_pos + 3 * (size_t)e; o[0] = rp.x; o[1] = rp.y; o[2] = rp.z;
        o = buf.char_root_rot + 4 * (size_t)e; o[0] = rq.x; o[1] = rq.y; o[2] = rq.z; o[3] = rq.w;
        o = buf.char_root_vel + 3 * (size_t)e; o[0] = rv.x; o[1] = rv.y; o[2] = rv.z;
        o = buf.char_root_ang_vel + 3 * (size_t)e; o[0] = rw.x; o[1] = rw.y; o[2] = rw.z;
    }
#pragma unroll
    for (int k = 0; k < CO_MAXLEN; ++k) {
        if (k < my_len) {
            const int b = bid[k], di = did[k];
            float *dp = buf.char_dof_pos + (size_t)Mg.D * e, *dv = buf.char_dof_vel + (size_t)Mg.D * e;
            if (jty[k] == DJ_SPHERICAL) {
                const v3 ex = qlog(jq[k]);
                dp[di] = ex.x; dp[di + 1] = ex.y; dp[di + 2] = ex.z; dv[di] = qd[k].x; dv[di + 1] = qd[k].y; dv[di + 2] = qd[k].z;
            } else if (jty[k] == DJ_HINGE) { dp[di] = hang[k]; dv[di] = qd[k].x; }
            float *cf = buf.contact_forces + 3 * ((size_t)e * B + b);
            cf[0] = fcon[k].x; cf[1] = fcon[k].y; cf[2] = fcon[k].z;
        }
    }
}
#endif // __HIPCC__

} // namespace parcdyn

#pragma clang fp contract(off)
