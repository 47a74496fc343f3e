// parc_env.hip — gfx950 kernels + the C-ABI of include/parc_env.h.
//
// One wavefront (64 lanes) owns one environment per iteration of a grid-stride loop:
//   * the joint hierarchy / per-body tables are staged once per wave into LDS;
//   * the 8 skeletons of a step (character, reference, 6 look-ahead targets) are 8 rows of 16 lanes:
//     lane (row, i) produces quaternion i of that row (slerp of two 512-byte frame records, or
//     dof->quat for the character) and its tan-norm observation in the same instruction stream;
//   * FK runs one root-to-leaf chain per lane (8 rows x 8 chains), no cross-lane traffic;
//   * the 441 height rays read a (2r+1)^2 terrain tile staged in LDS;
//   * the 1312-float observation row is assembled in LDS and streamed out as 16-byte stores.
// No MFMA: the work is quaternion algebra and gathers.  HBM traffic per env-step is the state read
// (456 B + 24 B bookkeeping) and the observation/reward write (5.3 KB); motion records and the terrain
// grid stay cache resident.
//
// Reference semantics: file:line citations relative to /root/reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/parc_env.h"
#include "parc_math.hpp"
#include "parc_dynamics.hpp"
#include "parc_dynamics_coop.hpp"
#include "parc_dynamics_wave.hpp"

using namespace parc;

// ------------------------------------------------------------------------------------------------
// device-side tables
// ------------------------------------------------------------------------------------------------
#define REC_F4 32            // one frame record = 32 float4 = 512 B
#define REC_Q_POS 15         // float4 #15 = root position
#define REC_Q_CONTACT 16     // float4 #16..19 = contacts
#define REC_Q_VEL 20         // float4 #20 = root_vel, #21 = root_ang_vel, #22.. = dof_vel

struct HierTables { // joint hierarchy.  The members up to `parent` are what every wave of k_env_post stages into LDS
    float lt[16][3];
    float lr[16][4];
    int fk_paths[PARC_MAX_FK_PATHS][PARC_MAX_FK_DEPTH];
    int key_ids[8];
    int key_slot[16]; // body -> index into key_ids, or -1
    int parent[16];   // ---- not staged (read from global memory where needed) ----
    int jtype[16];
    int dof_idx[16];
    float axis[16][4];
};
#define HIER_STAGED_WORDS ((int)(offsetof(HierTables, parent) / 4))

#define TILE_MAX_CELLS 320 // the terrain tile aliases the FK scratch (80 float4)

struct DevTables { // global memory; `h` is staged into LDS, the rest is read once per env
    HierTables h;
    float tstep[8];          // control_dt * tar_obs_steps (fp32 product, mgdm_dm_util.py:232)
    float joint_err_w[16];
    float dof_err_w[PARC_MAX_DOFS];
    float contact_w[16];
    float pose_term_dist[16];
};

struct MotionMeta { // 32 B
    int start, nframes;
    float length;
    int loop;
    float dx, dy, dz;
    float fps;
};

struct StepParams {
    int N, B, J, D, K, S, R, M, T;
    int lds_wave_floats; // dynamic LDS per wave of k_env_post
    int global_obs, off_char; // read by the OBSVAR instantiations of k_env_post only
    int lr_identity;          // every joint's fixed local rotation is the identity (the humanoid): lr (x) q = q, no quaternion product in the row phase
    int obs_tar, obs_contact; // OBSVAR only: 0 = the target block (+ target contacts) / the contact blocks and the reward's contact term are off
    int obs_dim, off_dofvel, off_key, off_tar, tar_w, off_tarc, off_cc, off_hf;
    float dt_f, episode_length, min_obs_h, max_obs_h;
    float pose_w, vel_w, root_pos_w, root_vel_w, key_pos_w;
    float root_pos_term_sq, root_rot_term;
    int early_term, pose_term, track_root, track_root_h, tracking, body_pos_from_fk, never_done;
    unsigned fall_mask;   // contact_bodies != []: bit b = body b may touch the ground (0 = the fall rule is off)
    float term_h;
    // terrain
    const float *hf; int X, Y; float min_x, min_y, dx, dy; int tile_r;
    float rdx, rdy;      // correctly rounded 1/dx, 1/dy (host): the ray loop divides by multiply + one exact correction
    unsigned tile_mul;   // idx / (2 tile_r + 1) == (idx * tile_mul) >> 16 for every tile cell (checked on the host)
    float tstep[8];      // control_dt * tar_obs_steps (fp32 product, mgdm_dm_util.py:232)
    // tables
    const float4 *records; const MotionMeta *meta; const float *motion_offsets; const float *env_offsets;
    const float *ray_points; const DevTables *tables;
    // curriculum hand-off
    unsigned char *ema_code;
    unsigned *stamp_out; // diagnostic builds only
    float4 *prep;        // [N][16]: slot 0 = (cos h, sin h, hinv.z, hinv.w), slots 1..B-1 = character joint quats
    ParcEnvBuffers buf;
};

struct Blend { int i0, i1; float b; };

// motion_lib.py:425-438 (+ calc_phase :520)
__device__ __forceinline__ Blend frame_blend(const MotionMeta &m, float t) {
    float phase = t / m.length;
    if (m.loop == PARC_LOOP_WRAP) phase = phase - floorf(phase);
    phase = fminf(fmaxf(phase, 0.f), 1.f);
    float pf = phase * (float)(m.nframes - 1);
    int f0 = (int)pf;                       // .long(): truncation
    f0 = max(0, min(f0, m.nframes - 1));    // memory safety only (no-op for finite phase)
    int f1 = min(f0 + 1, m.nframes - 1);
    Blend r;
    r.b = pf - (float)f0;
    r.i0 = f0 + m.start;
    r.i1 = f1 + m.start;
    return r;
}

// Joint.dof_to_rot kin_char_model.py:61-81
__device__ __forceinline__ Q4 joint_dof_to_rot(int type, const float *axis4, const float *dof) {
    if (type == PARC_JOINT_HINGE) return axis_angle_to_quat(mk3(axis4[0], axis4[1], axis4[2]), dof[0]);
    if (type == PARC_JOINT_SPHERICAL) return exp_map_to_quat(mk3(dof[0], dof[1], dof[2]));
    return mk4(0.f, 0.f, 0.f, 1.f);
}

// Joint.rot_to_dof kin_char_model.py:83-105
__device__ __forceinline__ void joint_rot_to_dof(int type, const float *axis4, Q4 q, float *out) {
    if (type == PARC_JOINT_HINGE) {
        V3 axis; float angle;
        quat_to_axis_angle(q, axis, angle);
        float dot = axis4[0] * axis.x + axis4[1] * axis.y + axis4[2] * axis.z;
        if (dot < 0.f) angle = angle * -1.f;
        out[0] = angle;
    } else if (type == PARC_JOINT_SPHERICAL) {
        V3 e = quat_to_exp_map(q);
        out[0] = e.x; out[1] = e.y; out[2] = e.z;
    }
}

// Same value as cell_index (IEEE division), for a divisor whose correctly rounded reciprocal rd is known: q0 = RN(x*rd),
// the residual x - d*q0 is exact in one fma, q = RN(q0 + r*rd) is the correctly rounded quotient (Markstein's
// reciprocal-with-correction division; checked exhaustively against x/d on the CPU by tests/test_host_cpu.py for the
// grid spacings in use).  3 instructions instead of the ~13 of the general sequence; used by the 441-ray loop.
__device__ __forceinline__ float cell_float_rcp(float p, float mn, float d, float rd) { // the cell index while still a float
    const float x = p - mn;
    const float q0 = x * rd;
    const float r = fmaf(-d, q0, x);
    return rintf(fmaf(r, rd, q0));
}
__device__ __forceinline__ int cell_index_rcp(float p, float mn, float d, float rd) {
    float f = cell_float_rcp(p, mn, d, rd);
    f = fminf(fmaxf(f, -1.0e9f), 1.0e9f);
    return (int)f;
}

// terrain_util.py:146-152 — unclamped nearest cell (round half to even)
__device__ __forceinline__ int cell_index(float p, float mn, float d) {
    float f = rintf((p - mn) / d);
    f = fminf(fmaxf(f, -1.0e9f), 1.0e9f);
    return (int)f;
}

// ------------------------------------------------------------------------------------------------
// k_dynamics: one THREAD per env advances the articulated character by one control step (nsub solver
// substeps) — clip action -> PD targets -> ABA with implicit drives and implicit cell-column contact
// (parc_dynamics.hpp).  Replaces gym.set_dof_position_target_tensor + gym.simulate x sim_steps + refresh_*
// (ig_char_env.py:488-497, ig_env.py:359-389).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dynamics(const parcdyn::DynModel *__restrict__ M, parcdyn::DynTerrain T, ParcEnvBuffers buf,
                                                 const float *__restrict__ action, const float *__restrict__ env_off, int N) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    parcdyn::DynState S;
    S.root_pos = buf.char_root_pos + 3 * (size_t)e; S.root_rot = buf.char_root_rot + 4 * (size_t)e;
    S.root_vel = buf.char_root_vel + 3 * (size_t)e; S.root_ang_vel = buf.char_root_ang_vel + 3 * (size_t)e;
    S.dof_pos = buf.char_dof_pos + (size_t)M->D * e; S.dof_vel = buf.char_dof_vel + (size_t)M->D * e;
    S.contact_force = buf.contact_forces + (size_t)3 * M->B * e; S.body_pos = nullptr;
    parcdyn::dyn_control_step(*M, T, S, action + (size_t)M->D * e, env_off + 3 * (size_t)e);
}

// ------------------------------------------------------------------------------------------------
// k_env_prep: the work that is scalar per env and transcendental-heavy — heading, inverse heading quaternion,
// cos/sin of the heading for the ray fan, and dof -> joint quaternion of the character (kin_char_model.py:586;
// torch_util.py:502-530).  In the wave-per-env kernel these would run with 15 of 64 lanes (measured: +35 us at 65 536
// envs when folded in, against the 22 us of this kernel); here all lanes are busy.  Output: a 256-byte record per env
// (L2 / Infinity-Cache resident between the two launches).  The kernels that WRITE the state form the same record with the
// same device functions (k_dynamics_wave after a step, k_reset_with after a reset); this kernel runs when the caller owns
// the state (parc_env_compute_obs, physics off) or one of the fallback dynamics kernels stepped it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_env_prep(const StepParams P, const int64_t *__restrict__ env_ids,
                                                 const int *__restrict__ env_ids32, const int *__restrict__ count_dev, int count) {
    // 16 lanes per env (4 envs per wave): lane 0 forms the heading terms, lane j the quaternion of joint j.  One
    // transcendental chain per lane instead of fifteen per thread: the same instruction count at 65 536 envs, a 10x shorter
    // critical path at the few thousand envs of a multi-GPU shard.
    if (count_dev) count = *count_dev;
    const int it = blockIdx.x * 4 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    if (it >= count || j >= P.B) return;
    const int e = env_ids ? (int)env_ids[it] : (env_ids32 ? env_ids32[it] : it);
    const DevTables *__restrict__ T = P.tables;
    float4 *out = P.prep + (size_t)e * 16;
    if (j == 0) {
        const float4 rr = *(const float4 *)(P.buf.char_root_rot + 4 * (size_t)e);
        const float heading = calc_heading(rr);
        const Q4 hinv = heading_quat_inv(heading);
        out[0] = make_float4(cosf(heading), sinf(heading), hinv.z, hinv.w);
    } else {
        const float *dof = P.buf.char_dof_pos + (size_t)e * P.D;
        const int ty = T->h.jtype[j], di = T->h.dof_idx[j];
        Q4 q = mk4(0.f, 0.f, 0.f, 1.f);
        if (ty == PARC_JOINT_HINGE) q = axis_angle_to_quat(mk3(T->h.axis[j][0], T->h.axis[j][1], T->h.axis[j][2]), dof[di]);
        else if (ty == PARC_JOINT_SPHERICAL) q = exp_map_to_quat(mk3(dof[di], dof[di + 1], dof[di + 2]));
        out[j] = q;
    }
}

// ------------------------------------------------------------------------------------------------
// the step kernel: IGEnv._post_physics_step (ig_env.py:368-377), one env per wave, four envs per 256-thread workgroup
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (-DPARC_STAMPS): per-phase s_memtime stamps of lane 0 go to a debug buffer that no
// other code reads; the shipped library is built without it.
#ifdef PARC_STAMPS
#define STAMP(i) do { if (lane == 0) s_stamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#define MODE_STEP 0
#define MODE_OBS 1
#define ENVS_PER_BLOCK 4
// orders the LDS traffic of ONE wave (see k_env_post): a compiler-level fence, no instruction
#define WAVE_SYNC() asm volatile("" ::: "memory")
#define RAY_UNROLL 8

__device__ __forceinline__ float wave_sum(float v) { // butterfly over the 64 lanes; every lane gets the total
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// Sum within each 16-lane row with four DPP row rotations (no LDS crossbar traffic): every lane of a row ends up with its row's total.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = v + dpp_f32<0x128>(v); // row_ror:8
    v = v + dpp_f32<0x124>(v); // row_ror:4
    v = v + dpp_f32<0x122>(v); // row_ror:2
    v = v + dpp_f32<0x121>(v); // row_ror:1
    return v;
}
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// MIRROR = false: none of the optional outputs (the ref_* mirrors of the reference's tensors, ray_hfs, tracking_error) is bound --
// the training configuration.  Their pointers and null checks then leave the kernel (20+ SGPRs: the kernel is at the SGPR limit and every
// scalar spilled to a VGPR lane comes back as a VALU instruction).
// LOCALROOT = the config's `track_root: false`: the reward compares root rotation, root velocities and key positions in each character's
// own heading frame (convert_to_local, mgdm_dm_util.py:247-267).  An instantiation of its own so that the default one carries none of it.
// OBSVAR = a non-default observation layout, decided at run time inside these instantiations only: `global_obs: true` (P.global_obs: the
// character and target observations stay in the global frame -- no heading rotation, the targets' key offsets are not shifted by the root
// offset) and / or `global_root_height_obs: true` (P.off_char = 1: the root height in front of the character block).  Instantiated with
// MIRROR = true only (off the default path).
template <int MODE, bool MIRROR, bool LOCALROOT = false, bool OBSVAR = false>
__global__ __launch_bounds__(256, 5) void k_env_post(const StepParams P, const int64_t *__restrict__ env_ids,
                                                 const int *__restrict__ env_ids32, const int *__restrict__ count_dev, int count,
                                                 unsigned long long *bump_calls) {
    // P arrives by value in the kernarg segment: its pointers are then known to be global (global_load / global_store
    // instead of flat_*, which would also tie up the LDS wait counter), its scalars are scalar loads where they are used
    // A workgroup is four waves = four envs (ENVS_PER_BLOCK).  Everything up to the reward is wave-private: each wave owns slice `wv` of
    // the LDS arrays and orders its own LDS traffic with WAVE_SYNC (LDS executes a wave's instructions in order; only the compiler has
    // to be kept from moving a read above the write of another lane).  The joint hierarchy is staged by every wave with the same values.
    extern __shared__ __align__(16) float s_obs_all[]; // per wave: staged observation prefix [0, off_tarc) (+ the body rotations of the tracking error)
    __shared__ int s_tab_raw[HIER_STAGED_WORDS];
    const HierTables &s_tab = *reinterpret_cast<const HierTables *>(s_tab_raw); // only the staged members are touched through it
    __shared__ float4 s_q_all[ENVS_PER_BLOCK][8][16];   // row r: quats 0..14 (0 = root), slot 15 = root position.  Target rows (r >= 2) hold lr (x) q for the joints
    __shared__ float4 s_lq_all[ENVS_PER_BLOCK][2][16];  // rows 0, 1: lr (x) q of the joints (s_q keeps their raw quaternions for the reward)
    __shared__ float4 s_fk_all[ENVS_PER_BLOCK][80];     // FK positions: rows 0,1 all bodies [r*16+b]; target rows key slots [32+(r-2)*8+k]
    // (LDS budget: 864 B of tables + 4 x (4 416 B static + 3 072 B staged row) = 30.1 KB per workgroup: 5 workgroups = 5 waves per SIMD
    // on a CU's 160 KB)
    __shared__ float s_cdofv_all[ENVS_PER_BLOCK][PARC_MAX_DOFS];
    __shared__ float4 s_refvel_all[ENVS_PER_BLOCK][12]; // record float4 #20..: root_vel, root_ang_vel, dof_vel
    __shared__ float s_refct_all[ENVS_PER_BLOCK][16];
    __shared__ float s_cfn_all[ENVS_PER_BLOCK][16];
    __shared__ float s_sc_all[ENVS_PER_BLOCK][16];      // per-env scalars handed to the wave that forms the rewards (MODE_STEP)
#ifdef PARC_STAMPS
    __shared__ unsigned long long s_stamp_all[ENVS_PER_BLOCK][16];
#endif
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool GLOBALOBS = OBSVAR && P.global_obs != 0; // a compile-time false in the default instantiations
    const int oc = OBSVAR ? P.off_char : 0;              // offset of the character block (1 when the root height leads the row)
    const bool TAROBS = !OBSVAR || P.obs_tar != 0;       // enable_tar_obs (ig_parkour_env.py:83): the look-ahead target block exists
    const bool CONTACTOBS = !OBSVAR || P.obs_contact != 0; // use_contact_info (:72): contact blocks of the observation, contact term of the reward
    float *s_obs = s_obs_all + (size_t)wv * P.lds_wave_floats;
    float4 (*s_q)[16] = s_q_all[wv];
    float4 (*s_lq)[16] = s_lq_all[wv];
    float4 *s_fk = s_fk_all[wv];
    float4 *s_cbp = s_lq[0];        // simulator rigid-body positions of the character: written once the FK chains are through with s_lq
    float *s_cdofv = s_cdofv_all[wv];
    float4 *s_refvel = s_refvel_all[wv];
    float *s_refct = s_refct_all[wv], *s_cfn = s_cfn_all[wv];
#ifdef PARC_STAMPS
    unsigned long long *s_stamp = s_stamp_all[wv];
#endif
    float *s_tile = (float *)s_fk;  // the terrain tile is dead before FK writes s_fk
    float4 (*s_br)[16] = reinterpret_cast<float4 (*)[16]>(s_obs + ((P.off_tarc + 3) & ~3)); // body rotations of char/ref: only allocated
                                                                                            // (behind the staged prefix) when the tracking error is reported

    // the observation pass of a sampling reset closes the call: every block of the reset kernel has read the Philox call index by now
    if (MODE == MODE_OBS && bump_calls && blockIdx.x == 0 && threadIdx.x == 0) *bump_calls += 1ull;
    if (count_dev) count = *count_dev; // device-side list (reset_done): no host round trip
    const int it = blockIdx.x * ENVS_PER_BLOCK + wv;
    if (it >= count) return;
    int e = env_ids ? (int)env_ids[it] : (env_ids32 ? env_ids32[it] : it);
    e = __builtin_amdgcn_readfirstlane(e);
    STAMP(0);

    const int B = P.B, D = P.D, S = P.S, K = P.K, J = P.J;
    const DevTables *__restrict__ T = P.tables;

    float *orow = P.buf.obs + (size_t)e * P.obs_dim;

    // ================= prefetch: every global load of this env is issued before any is consumed =================
    // (a) loads that only need the env id go first, so they overlap the scalar bookkeeping chain below
    constexpr int HW = HIER_STAGED_WORDS;
    constexpr int HN = (HW + 63) / 64;
    int tabreg[HN];
#pragma unroll
    for (int i = 0; i < HN; ++i) { const int w = lane + 64 * i; tabreg[i] = w < HW ? ((const int *)&T->h)[w] : 0; }
    float dofv = 0.f;
    if (lane < D) dofv = P.buf.char_dof_vel[(size_t)e * D + lane];
    float aux0 = 0.f, aux1 = 0.f, aux2 = 0.f; // lanes 32..32+B: contact force; 30/31: root (ang) vel
    {
        const float *src = nullptr;
        if (lane >= 32 && lane < 32 + B) src = P.buf.contact_forces + 3 * ((size_t)e * B + (lane - 32));
        else if (lane == 30) src = P.buf.char_root_vel + 3 * (size_t)e;
        else if (lane == 31) src = P.buf.char_root_ang_vel + 3 * (size_t)e;
        if (src) { aux0 = src[0]; aux1 = src[1]; aux2 = src[2]; }
    }
    const int qi = lane & 15;
    float4 prepq = make_float4(0.f, 0.f, 0.f, 1.f);
    if (lane < 16) prepq = P.prep[(size_t)e * 16 + lane]; // slot 0 heading terms, 1..B-1 character joint quats (k_env_prep)
    float2 rayp[RAY_UNROLL];
#pragma unroll
    for (int i = 0; i < RAY_UNROLL; ++i) {
        const int r = lane + 64 * i;
        rayp[i] = r < P.R ? ((const float2 *)P.ray_points)[r] : make_float2(0.f, 0.f);
    }

    // (b) bookkeeping + time (ig_env.py:391-394, dm_env.py:547-552): uniform, scalar loads
    const int mid = P.buf.motion_ids[e];
    const int tid = P.buf.terrain_ids[e];
    const float toff = P.buf.time_offsets[e];
    int ts = P.buf.timestep[e];
    if (MODE == MODE_STEP) ts += 1;
    const float time = P.dt_f * (float)ts;
    const float mt = time + toff;
    const MotionMeta meta = P.meta[mid];
    const float eox = P.env_offsets[3 * e + 0], eoy = P.env_offsets[3 * e + 1], eoz = P.env_offsets[3 * e + 2];
    const float *mo = P.motion_offsets + 2 * ((size_t)mid * P.T + tid);
    const float offx = mo[0] - eox, offy = mo[1] - eoy; // dm_env.py:556
    const float *crp = P.buf.char_root_pos + 3 * (size_t)e;
    const float *crr = P.buf.char_root_rot + 4 * (size_t)e;
    const V3 root_pos = mk3(crp[0], crp[1], crp[2]);
    const Q4 root_rot = mk4(crr[0], crr[1], crr[2], crr[3]);
    const float gx = root_pos.x + eox, gy = root_pos.y + eoy, gz = root_pos.z + eoz; // ig_parkour_env.py:522

    // (c) reference motion: rows 1..1+S (row = pass*4 + lane/16), two 512-byte frame records per sample.  The frame pair and blend factor
    // of a sample (motion_lib.py:425-438) are formed once, by lane (sample & 7): sample 0 = the reference at t, sample s = look-ahead
    // target s; the row / contact / velocity lanes fetch theirs with a lane shuffle instead of evaluating the blend four times per lane.
    Blend myb;
    {
        const int sid = lane & 7;
        float t = mt;
#pragma unroll
        for (int q = 0; q < PARC_MAX_TAR_STEPS; ++q) t = (sid - 1 == q) ? mt + P.tstep[q] : t;
        myb = frame_blend(meta, t);
    }
    const int last_frame = meta.start + meta.nframes - 1;
    float4 fA[2], fB[2];
    Blend bl[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 4 + (lane >> 4);
        const int src = max(r - 1, 0);
        bl[p].b = __shfl(myb.b, src, 64);
        bl[p].i0 = __shfl(myb.i0, src, 64);
        bl[p].i1 = min(bl[p].i0 + 1, last_frame); // = frame_blend's i1
        fA[p] = make_float4(0.f, 0.f, 0.f, 1.f); fB[p] = fA[p];
        if (r >= 1 && r < 2 + S) {
            fA[p] = P.records[(size_t)bl[p].i0 * REC_F4 + qi];
            fB[p] = P.records[(size_t)bl[p].i1 * REC_F4 + qi];
        }
    }
    // (d) contacts of the 1+S samples (lanes < 4(1+S)) / velocity block of sample 0 (lanes 32..)
    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
    float cblend = 0.f;
    const int nvel = 2 + (D + 3) / 4;
    {
        const int src = lane < 4 * (1 + S) ? lane >> 2 : 0;
        const float b2b = __shfl(myb.b, src, 64);
        const int b2i0 = __shfl(myb.i0, src, 64), b2i1 = min(b2i0 + 1, last_frame);
        if (lane < 4 * (1 + S)) {
            const int c = lane & 3;
            cblend = b2b;
            cA = P.records[(size_t)b2i0 * REC_F4 + REC_Q_CONTACT + c];
            cB = P.records[(size_t)b2i1 * REC_F4 + REC_Q_CONTACT + c];
        } else if (lane >= 32 && lane < 32 + nvel) { // velocities come from frame idx0 un-interpolated (:103-109)
            cA = P.records[(size_t)b2i0 * REC_F4 + REC_Q_VEL + (lane - 32)];
        }
    }
    // (e) terrain tile cells
    const int tr = P.tile_r, TW = 2 * tr + 1, ncell = tr >= 0 ? TW * TW : 0;
    // (the origin is kept within +-2^22 cells so that it and the tile bounds are exact as floats: the ray loop clamps in float)
    const int ox = min(max(cell_index(gx, P.min_x, P.dx) - tr, -(1 << 22)), 1 << 22), oy = min(max(cell_index(gy, P.min_y, P.dy) - tr, -(1 << 22)), 1 << 22);
    float tilev[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        tilev[i] = 0.f;
        if (idx < ncell) {
            const int a = (int)(((unsigned)idx * P.tile_mul) >> 16), bq = idx - a * TW;
            const int cx = min(max(ox + a, 0), P.X - 1), cy = min(max(oy + bq, 0), P.Y - 1);
            tilev[i] = P.hf[(size_t)cx * P.Y + cy];
        }
    }

    // ================= heading terms from k_env_prep (uniform), LDS fills =================
    const float ch = __shfl(prepq.x, 0, 64), sh = __shfl(prepq.y, 0, 64);
    const Q4 hinv = mk4(0.f, 0.f, __shfl(prepq.z, 0, 64), __shfl(prepq.w, 0, 64)); // axis_angle_to_quat(z, -heading): x = y = 0 exactly
#pragma unroll
    for (int i = 0; i < HN; ++i) { const int w = lane + 64 * i; if (w < HW) s_tab_raw[w] = tabreg[i]; }
    if (lane < D) {
        s_cdofv[lane] = dofv;
        s_obs[P.off_dofvel + lane] = dofv;
    }
    if (OBSVAR && oc && lane == 29) s_obs[0] = root_pos.z; // root_height_obs (ig_char_env.py:620-622)
    if (lane == 30 || lane == 31) { // root velocities in the heading frame (ig_char_env.py:593-597)
        const V3 r = GLOBALOBS ? mk3(aux0, aux1, aux2) : quat_rotate(hinv, mk3(aux0, aux1, aux2));
        const int o = oc + (lane == 30 ? 6 : 9);
        s_obs[o + 0] = r.x; s_obs[o + 1] = r.y; s_obs[o + 2] = r.z;
    }
    if (lane >= 32 && lane < 32 + B) { // contact flags + clamped force norms (ig_parkour_env.py:655-662, mgdm_dm_util.py:505-508)
        const float n = norm3(mk3(aux0, aux1, aux2));
        if (CONTACTOBS) orow[P.off_cc + (lane - 32)] = n > 1e-5f ? 1.f : 0.f;
        s_cfn[lane - 32] = fminf(n, 1.0f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) { const int idx = lane + 64 * i; if (idx < ncell) s_tile[idx] = tilev[i]; }
    WAVE_SYNC();
    STAMP(1);

    // ================= height rays (mgdm_dm_util.py:128-145; terrain_util.py:146-156) =================
    // Two loops, one per source of the heights, so that the common one reads the LDS tile with plain ds_read (a pointer that may
    // be LDS or global becomes a flat load, whose wait also covers the stores of the previous ray).
    {
        float *hrow = orow + P.off_hf;
        float *hmirror = MIRROR && P.buf.ray_hfs ? P.buf.ray_hfs + (size_t)e * P.R : nullptr;
        for (int base = 0; base < P.R; base += 64 * RAY_UNROLL) {
            if (base > 0) {
#pragma unroll
                for (int i = 0; i < RAY_UNROLL; ++i) { const int r = base + lane + 64 * i; rayp[i] = r < P.R ? ((const float2 *)P.ray_points)[r] : make_float2(0.f, 0.f); }
            }
            const float tlox = (float)ox, thix = (float)(ox + TW - 1), tloy = (float)oy, thiy = (float)(oy + TW - 1);
            if (tr >= 0) { // the tile radius covers the whole fan by construction (parc_env_load_terrain): clamp, no fallback
                // every tile index is formed before the first store: a later use of a loaded ray point would have to wait for
                // the stores issued in between (one counter orders loads and stores).  Lanes past the end of the fan carry the
                // point (0, 0), a valid cell; only their stores are masked.
                int tix[RAY_UNROLL];
#pragma unroll
                for (int i = 0; i < RAY_UNROLL; ++i) {
                    tix[i] = 0;
                    if (base + 64 * i < P.R) { // uniform
                        const float px = (rayp[i].x * ch - rayp[i].y * sh) + gx; // rotate_2d_vec torch_util.py:651
                        const float py = (rayp[i].x * sh + rayp[i].y * ch) + gy;
                        // nearest cell (cell_index_rcp), clamped to the tile while still a float: the bounds are small integers, so
                        // clamp-then-convert equals convert-then-clamp and the guard against a wild float comes for free
                        const int a = (int)fminf(fmaxf(cell_float_rcp(px, P.min_x, P.dx, P.rdx), tlox), thix) - ox;
                        const int bq = (int)fminf(fmaxf(cell_float_rcp(py, P.min_y, P.dy, P.rdy), tloy), thiy) - oy;
                        tix[i] = a * TW + bq;
                    }
                    asm volatile("" : "+v"(tix[i])); // keep the index arithmetic here (not sunk below the stores)
                }
                float hv[RAY_UNROLL];
#pragma unroll
                for (int i = 0; i < RAY_UNROLL; ++i) hv[i] = s_tile[tix[i]]; // one batch of LDS reads, one wait
#pragma unroll
                for (int i = 0; i < RAY_UNROLL; ++i) {
                    const int r = base + lane + 64 * i;
                    if (r < P.R) {
                        float h = hv[i] - gz;
                        h = fminf(fmaxf(h, P.min_obs_h), P.max_obs_h);
                        hrow[r] = h; // 256 contiguous bytes per wave store: no staging needed
                        if (hmirror) hmirror[r] = h;
                    }
                }
            } else { // fan too wide for the LDS tile: direct gathers
#pragma unroll
                for (int i = 0; i < RAY_UNROLL; ++i) {
                    const int r = base + lane + 64 * i;
                    if (r < P.R) {
                        const float px = (rayp[i].x * ch - rayp[i].y * sh) + gx;
                        const float py = (rayp[i].x * sh + rayp[i].y * ch) + gy;
                        const int ix = cell_index_rcp(px, P.min_x, P.dx, P.rdx), iy = cell_index_rcp(py, P.min_y, P.dy, P.rdy);
                        const int cx = min(max(ix, 0), P.X - 1), cy = min(max(iy, 0), P.Y - 1);
                        float h = P.hf[(size_t)cx * P.Y + cy] - gz;
                        h = fminf(fmaxf(h, P.min_obs_h), P.max_obs_h);
                        hrow[r] = h;
                        if (hmirror) hmirror[r] = h;
                    }
                }
            }
        }
    }
    STAMP(2);

    // ================= 8 rows x 16 lanes: quaternions of char / ref / targets + tan-norm observations =================
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 4 + (lane >> 4);
        const int i = qi;
        if (r < 2 + S) {
            float4 res = make_float4(0.f, 0.f, 0.f, 1.f);
            if (r == 0) { // character: dof -> quat (kin_char_model.py:586)
                if (i == 0) res = root_rot;
                else if (i == 15) res = make_float4(root_pos.x, root_pos.y, root_pos.z, 0.f);
                else if (i < B) res = prepq;
            } else { // reference motion sample (motion_lib.py:94-128)
                const float4 A = fA[p], Bv = fB[p];
                const float bb = bl[p].b;
                if (i == 15) {
                    const float a = 1.0f - bb;
                    res.x = a * A.x + bb * Bv.x;
                    res.y = a * A.y + bb * Bv.y;
                    res.z = a * A.z + bb * Bv.z;
                    res.w = 0.f;
                    if (meta.loop == PARC_LOOP_WRAP) { // _calc_loop_offset :440
                        float t = mt;
#pragma unroll
                        for (int q = 0; q < PARC_MAX_TAR_STEPS; ++q) t = (r - 2 == q) ? mt + P.tstep[q] : t;
                        const float ph = floorf(t / meta.length);
                        res.x = res.x + ph * meta.dx; res.y = res.y + ph * meta.dy; res.z = res.z + ph * meta.dz;
                    }
                    res.x = res.x + offx; // _move_to_motion_terrain dm_env.py:554
                    res.y = res.y + offy;
                } else if (i < B) {
                    res = slerp_rr(A, Bv, bb);
                }
            }
            if (i >= 1 && i < B) { // the FK chains multiply parent (x) (lr (x) q): the inner product is formed here, once per joint,
                                   // instead of once per chain level (same operations, same order)
                const Q4 lq = P.lr_identity ? res : quat_mul(mk4(s_tab.lr[i][0], s_tab.lr[i][1], s_tab.lr[i][2], s_tab.lr[i][3]), res);
                if (r < 2) { s_lq[r][i] = lq; s_q[r][i] = res; } else s_q[r][i] = lq;
            } else {
                s_q[r][i] = res;
            }
            if (r == 1) { // optional mirrors of the reference's ref_* tensors
                if (MIRROR && i == 0 && P.buf.ref_root_rot) *(float4 *)(P.buf.ref_root_rot + 4 * (size_t)e) = res;
                if (MIRROR && i >= 1 && i < B && P.buf.ref_joint_rot) *(float4 *)(P.buf.ref_joint_rot + 4 * ((size_t)e * J + i - 1)) = res;
                if (MIRROR && i == 15 && P.buf.ref_root_pos) {
                    float *o = P.buf.ref_root_pos + 3 * (size_t)e;
                    o[0] = res.x; o[1] = res.y; o[2] = res.z;
                }
            } else if (r == 0 || TAROBS) { // observation pieces (ig_char_env.py:582, mgdm_dm_util.py:405)
                const int base = r == 0 ? 0 : P.off_tar + (r - 2) * P.tar_w;
                if (i < B) {
                    const Q4 qq = (i == 0 && !GLOBALOBS) ? quat_mul(hinv, res) : res;
                    float tn[6];
                    quat_to_tan_norm(qq, tn);
                    const int o = r == 0 ? oc + (i == 0 ? 0 : 12 + 6 * (i - 1)) : base + 3 + 6 * i;
#pragma unroll
                    for (int c = 0; c < 6; ++c) s_obs[o + c] = tn[c];
                } else if (i == 15 && r >= 2) {
                    const V3 rpd = mk3(res.x - root_pos.x, res.y - root_pos.y, res.z - root_pos.z);
                    const V3 rpo = GLOBALOBS ? rpd : quat_rotate(hinv, rpd);
                    s_obs[base + 0] = rpo.x; s_obs[base + 1] = rpo.y; s_obs[base + 2] = rpo.z;
                }
            }
        }
    }
    // ---- contacts of the 1+S samples and the velocity block of sample 0 ---------------------------------
    if (lane < 4 * (1 + S)) {
        const int s = lane >> 2, c = lane & 3;
        const float b = cblend, a = 1.0f - b;
        const float v[4] = {a * cA.x + b * cB.x, a * cA.y + b * cB.y, a * cA.z + b * cB.z, a * cA.w + b * cB.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bd = 4 * c + k;
            if (bd < B) {
                if (s == 0) {
                    s_refct[bd] = v[k];
                    if (MIRROR && P.buf.ref_contacts) P.buf.ref_contacts[(size_t)e * B + bd] = v[k];
                } else if (TAROBS && CONTACTOBS) {
                    orow[P.off_tarc + (s - 1) * B + bd] = v[k];
                }
            }
        }
    } else if (lane >= 32 && lane < 32 + nvel) {
        const int c = lane - 32;
        s_refvel[c] = cA;
        if (MIRROR && c == 0 && P.buf.ref_root_vel) { float *o = P.buf.ref_root_vel + 3 * (size_t)e; o[0] = cA.x; o[1] = cA.y; o[2] = cA.z; }
        if (MIRROR && c == 1 && P.buf.ref_root_ang_vel) { float *o = P.buf.ref_root_ang_vel + 3 * (size_t)e; o[0] = cA.x; o[1] = cA.y; o[2] = cA.z; }
        if (MIRROR && c >= 2 && P.buf.ref_dof_vel) {
            const float vv[4] = {cA.x, cA.y, cA.z, cA.w};
            for (int k = 0; k < 4; ++k) { const int d = 4 * (c - 2) + k; if (d < D) P.buf.ref_dof_vel[(size_t)e * D + d] = vv[k]; }
        }
    }
    float bpx = 0.f, bpy = 0.f, bpz = 0.f; // caller-provided rigid-body positions (physics off): in flight during the FK chains
    if (!P.body_pos_from_fk && lane < B) {
        const float *src = P.buf.char_body_pos + 3 * ((size_t)e * B + lane);
        bpx = src[0]; bpy = src[1]; bpz = src[2];
    }
    WAVE_SYNC();
    STAMP(3);

    // ================= FK: row k = lane>>3, root-to-leaf chain c = lane&7 (kin_char_model.py:617-649) =================
    {
        const int k = lane >> 3, c = lane & 7;
        if (k < 2 + S) {
            Q4 prot = s_q[k][0];
            const float4 *jq = k < 2 ? s_lq[k] : s_q[k];
            const float4 pp = s_q[k][15];
            V3 ppos = mk3(pp.x, pp.y, pp.z);
            if (c == 0 && k < 2) {
                s_fk[k * 16] = make_float4(ppos.x, ppos.y, ppos.z, 0.f);
                if (MIRROR && P.tracking) s_br[k][0] = prot;
            }
#pragma unroll 1
            for (int d = 0; d < PARC_MAX_FK_DEPTH; ++d) {
                const int b = s_tab.fk_paths[c][d];
                if (b < 0) break;
                const V3 wt = quat_rotate(prot, mk3(s_tab.lt[b][0], s_tab.lt[b][1], s_tab.lt[b][2]));
                ppos = mk3(ppos.x + wt.x, ppos.y + wt.y, ppos.z + wt.z);
                prot = quat_mul(prot, jq[b]);
                if (k < 2) {
                    s_fk[k * 16 + b] = make_float4(ppos.x, ppos.y, ppos.z, 0.f);
                    if (MIRROR && P.tracking) s_br[k][b] = prot;
                } else {
                    const int slot = s_tab.key_slot[b];
                    if (slot >= 0) s_fk[32 + (k - 2) * 8 + slot] = make_float4(ppos.x, ppos.y, ppos.z, 0.f);
                }
            }
        }
    }
    WAVE_SYNC();
    if (lane < B) s_cbp[lane] = P.body_pos_from_fk ? s_fk[lane] : make_float4(bpx, bpy, bpz, 0.f);
    STAMP(4);

    // ================= key-body observations (ig_char_env.py:603-617, mgdm_dm_util.py:415-440) =================
    if (lane < 8 * (1 + S)) {
        const int row = lane >> 3, kk = lane & 7; // row 0 = char, 1.. = target rows 2..
        if (kk < K) {
            if (row == 0) {
                const float4 kp = s_fk[s_tab.key_ids[kk]];
                const V3 rd = mk3(kp.x - root_pos.x, kp.y - root_pos.y, kp.z - root_pos.z);
                const V3 rl = GLOBALOBS ? rd : quat_rotate(hinv, rd);
                const int o = P.off_key + 3 * kk;
                s_obs[o] = rl.x; s_obs[o + 1] = rl.y; s_obs[o + 2] = rl.z;
            } else if (TAROBS) {
                const int r = row + 1;
                const float4 kp = s_fk[32 + (r - 2) * 8 + s_tab.key_slot[s_tab.key_ids[kk]]], trp = s_q[r][15];
                const V3 rd = mk3(kp.x - trp.x, kp.y - trp.y, kp.z - trp.z);
                const int tb = P.off_tar + (r - 2) * P.tar_w, o = tb + 3 + 6 * B + 3 * kk; // [tb, tb + 3): the row's root offset, staged by the row phase
                if (GLOBALOBS) { // mgdm_dm_util.py:417: global key offsets are not shifted by the root offset
                    s_obs[o] = rd.x; s_obs[o + 1] = rd.y; s_obs[o + 2] = rd.z;
                } else {
                    const V3 rl = quat_rotate(hinv, rd);
                    s_obs[o] = rl.x + s_obs[tb]; s_obs[o + 1] = rl.y + s_obs[tb + 1]; s_obs[o + 2] = rl.z + s_obs[tb + 2];
                }
            }
        }
    }
    STAMP(5);

    if (MODE == MODE_STEP) {
        // ---- reward + done of the workgroup's four envs on ONE of its waves: 16 lanes per env (row j = env j, lane i of the row = one
        // joint / body / dof / term).  Every sum, exponential and comparison below is one instruction stream for four envs instead of one
        // per env on 1-16 of 64 lanes.  Each wave hands its per-env scalars over through s_sc; the quaternions, FK positions, body
        // positions, reference velocities and contact terms are already in its LDS slice.  The owner rotates with the block index so
        // that the extra work does not pile up on one SIMD.
        {
            float *sc = s_sc_all[wv];
            bool fc = false; // fall rule, first half (mgdm_dm_util.py:349-360): a contact-force component above 0.1 on a non-contact body
            if (P.fall_mask != 0u && lane >= 32 && lane < 32 + B && !((P.fall_mask >> (lane - 32)) & 1u))
                fc = fabsf(aux0) > 0.1f || fabsf(aux1) > 0.1f || fabsf(aux2) > 0.1f;
            const bool fcany = __ballot(fc) != 0ull;
            if (lane == 30) { sc[5] = aux0; sc[6] = aux1; sc[7] = aux2; }
            if (lane == 31) { sc[8] = aux0; sc[9] = aux1; sc[10] = aux2; }
            if (lane == 0) {
                sc[0] = time; sc[1] = mt; sc[2] = __int_as_float(ts); sc[3] = meta.length; sc[4] = __int_as_float(meta.loop);
                sc[11] = fcany ? 1.f : 0.f; sc[12] = eox; sc[13] = eoy; sc[14] = __int_as_float(e);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the four waves' LDS slices are complete (waves of a short last block have left)
        const int nact = min(ENVS_PER_BLOCK, count - (int)blockIdx.x * ENVS_PER_BLOCK);
        int owner = blockIdx.x & (ENVS_PER_BLOCK - 1);
        if (owner >= nact) owner = 0;
        if (wv == owner) {
            const int j = lane >> 4, i = lane & 15, rowbase = lane & 48;
            const bool valid = j < nact;
            const int jj = valid ? j : 0;
            const float4 (*q)[16] = s_q_all[jj];
            const float4 *fk = s_fk_all[jj];
            const float4 *cbp = s_lq_all[jj][0];
            const float *scj = s_sc_all[jj];
            const float4 rp4 = q[0][15], trp = q[1][15];
            // reward terms, one per lane (mgdm_dm_util.py:270-333, 498-518); every row sum is taken at the row's lane 0
            float dang = 0.f, vpose = 0.f;
            Q4 hinv_c = mk4(0.f, 0.f, 0.f, 1.f), hinv_r = hinv_c; // LOCALROOT: inverse heading of the character / of the reference
            if (LOCALROOT) {
                hinv_c = heading_quat_inv(calc_heading(q[0][0]));
                hinv_r = heading_quat_inv(calc_heading(q[1][0]));
            }
            if (i <= J) { // angle of ref (x) conj(char): joints on lanes < J, the root on lane J (slot 0 of both rows)
                const int slot = i < J ? 1 + i : 0;
                if (LOCALROOT) {
                    Q4 qa = q[0][slot], qb = q[1][slot];
                    if (i == J) { qa = quat_mul(hinv_c, qa); qb = quat_mul(hinv_r, qb); }
                    dang = quat_diff_angle(qa, qb);
                } else {
                    dang = quat_diff_angle(q[0][slot], q[1][slot]);
                }
                if (i < J) vpose = T->joint_err_w[i] * dang * dang;
            }
            float vs[4] = {0.f, 0.f, 0.f, 0.f}; // dof velocity terms, 16 dofs per pass (the sums pair up as the four 16-lane rows of a wave would)
#pragma unroll
            for (int pss = 0; pss < (PARC_MAX_DOFS + 15) / 16; ++pss) {
                if (16 * pss < D) { // uniform
                    const int d = 16 * pss + i;
                    float v = 0.f;
                    if (d < D) {
                        const float dv = ((const float *)s_refvel_all[jj])[8 + d] - s_cdofv_all[jj][d];
                        v = T->dof_err_w[d] * dv * dv;
                    }
                    vs[pss] = row_sum16(v);
                }
            }
            float vkey = 0.f;
            if (i < K) { // key positions: simulator bodies vs reference FK (ig_parkour_env.py:987)
                const int b = s_tab.key_ids[i];
                const float4 kp = cbp[b], tk = fk[16 + b];
                if (LOCALROOT) {
                    const V3 kt = quat_rotate(hinv_r, mk3(tk.x - trp.x, tk.y - trp.y, tk.z - trp.z));
                    const V3 kc = quat_rotate(hinv_c, mk3(kp.x - rp4.x, kp.y - rp4.y, kp.z - rp4.z));
                    const float dx = kt.x - kc.x, dy = kt.y - kc.y, dz = kt.z - kc.z;
                    vkey = dx * dx + dy * dy + dz * dz;
                } else {
                    const float dx = (tk.x - trp.x) - (kp.x - rp4.x);
                    const float dy = (tk.y - trp.y) - (kp.y - rp4.y);
                    const float dz = (tk.z - trp.z) - (kp.z - rp4.z);
                    vkey = dx * dx + dy * dy + dz * dz;
                }
            }
            float vct = 0.f;
            if (i < B) { // contact term
                const float tar = s_refct_all[jj][i], f = s_cfn_all[jj][i];
                float cr = -(1.0f - tar) * f;
                cr = cr + tar * f;
                vct = T->contact_w[i] * cr;
            }
            // ---- early termination (mgdm_dm_util.py:335-402) ----
            bool bad = false;
            if (i >= 1 && i < B) {
                const float4 bp = cbp[i], b0 = cbp[0], tp = fk[16 + i], t0 = fk[16];
                const float dx = (tp.x - t0.x) - (bp.x - b0.x);
                const float dy = (tp.y - t0.y) - (bp.y - b0.y);
                const float dz = (tp.z - t0.z) - (bp.z - b0.z);
                const float lim = T->pose_term_dist[i - 1];
                bad = (dx * dx + dy * dy + dz * dz) > lim * lim;
            }
            const bool pose_fail_any = ((unsigned)(__ballot(bad) >> rowbase) & 0xffffu) != 0u;
            // fall rule, second half: a non-contact body lower than termination_height above the terrain under it (global xy = env-local
            // + env offset, :147-152)
            bool fallen = false;
            if (P.fall_mask != 0u) { // uniform
                bool fh = false;
                if (i < B && !((P.fall_mask >> i) & 1u)) {
                    const float4 bp = cbp[i];
                    const int ix = min(max(cell_index(bp.x + scj[12], P.min_x, P.dx), 0), P.X - 1), iy = min(max(cell_index(bp.y + scj[13], P.min_y, P.dy), 0), P.Y - 1);
                    fh = bp.z < P.hf[(size_t)ix * P.Y + iy] + P.term_h;
                }
                fallen = scj[11] != 0.f && ((unsigned)(__ballot(fh) >> rowbase) & 0xffffu) != 0u;
            }
            const float root_rot_angle = __shfl(dang, rowbase + J, 64);
            const float pose_err = __shfl(row_sum16(vpose), rowbase, 64), csum = __shfl(row_sum16(vct), rowbase, 64), key_err = __shfl(row_sum16(vkey), rowbase, 64);
            const float vel_err = __shfl((vs[0] + vs[1]) + (vs[2] + vs[3]), rowbase, 64);

            // per-env tail: every lane of a row carries its env's values; the five exponentials run on lanes 0..4 of the row
            const float time_j = scj[0], mt_j = scj[1], mlen_j = scj[3];
            const int ts_j = __float_as_int(scj[2]), mloop_j = __float_as_int(scj[4]), e_j = __float_as_int(scj[14]);
            float rdx = trp.x - rp4.x, rdy = trp.y - rp4.y, rdz = trp.z - rp4.z;
            if (!P.track_root) { rdx = 0.f; rdy = 0.f; }
            if (!P.track_root_h) rdz = 0.f;
            const float root_pos_err = rdx * rdx + rdy * rdy + rdz * rdz;
            const float4 tv = s_refvel_all[jj][0], tav = s_refvel_all[jj][1];
            const float rre = root_rot_angle * root_rot_angle;
            float rve, rave;
            if (LOCALROOT) {
                const V3 tv3 = quat_rotate(hinv_r, mk3(tv.x, tv.y, tv.z)), tav3 = quat_rotate(hinv_r, mk3(tav.x, tav.y, tav.z));
                const V3 rv3 = quat_rotate(hinv_c, mk3(scj[5], scj[6], scj[7])), rav3 = quat_rotate(hinv_c, mk3(scj[8], scj[9], scj[10]));
                float d0 = tv3.x - rv3.x, d1 = tv3.y - rv3.y, d2 = tv3.z - rv3.z;
                rve = d0 * d0 + d1 * d1 + d2 * d2;
                d0 = tav3.x - rav3.x; d1 = tav3.y - rav3.y; d2 = tav3.z - rav3.z;
                rave = d0 * d0 + d1 * d1 + d2 * d2;
            } else {
                float d0 = tv.x - scj[5], d1 = tv.y - scj[6], d2 = tv.z - scj[7];
                rve = d0 * d0 + d1 * d1 + d2 * d2;
                d0 = tav.x - scj[8]; d1 = tav.y - scj[9]; d2 = tav.z - scj[10];
                rave = d0 * d0 + d1 * d1 + d2 * d2;
            }
            float earg = -0.25f * pose_err;
            earg = i == 1 ? -0.01f * vel_err : earg;
            earg = i == 2 ? -5.0f * (root_pos_err + 0.1f * rre) : earg;
            earg = i == 3 ? -1.0f * (rve + 0.1f * rave) : earg;
            earg = i == 4 ? -10.0f * key_err : earg;
            const float eval = expf(earg);
            const float pose_r = __shfl(eval, rowbase, 64), vel_r = __shfl(eval, rowbase + 1, 64), root_pose_r = __shfl(eval, rowbase + 2, 64),
                        root_vel_r = __shfl(eval, rowbase + 3, 64), key_pos_r = __shfl(eval, rowbase + 4, 64);
            float rew = P.pose_w * pose_r + P.vel_w * vel_r + P.root_pos_w * root_pose_r + P.root_vel_w * root_vel_r + P.key_pos_w * key_pos_r;
            const float contact_pen = (OBSVAR && P.obs_contact == 0) ? 0.f : csum / (float)B; // torch.mean over bodies (ig_parkour_env.py:1033); no term without use_contact_info (:1032)
            rew = rew + contact_pen;
            // done (compute_done + DeepMimicEnv.update_done dm_env.py:628-665)
            int done = PARC_DONE_NULL;
            if (time_j >= P.episode_length) done = PARC_DONE_TIME;
            if (P.early_term) {
                bool failed = fallen;
                if (P.pose_term) {
                    bool pf = pose_fail_any;
                    if (P.track_root) {
                        const float4 b0 = cbp[0], t0 = fk[16];
                        const float ex = b0.x - t0.x, ey = b0.y - t0.y, ez = b0.z - t0.z;
                        pf = pf || (ex * ex + ey * ey + ez * ez) > P.root_pos_term_sq;
                        pf = pf || fabsf(root_rot_angle) > P.root_rot_term;
                    }
                    failed = failed || pf;
                }
                if (!(time_j > 1e-5f)) failed = false;
                if (failed) done = PARC_DONE_FAIL;
            }
            const bool motion_end = (mt_j >= mlen_j) && (mloop_j != PARC_LOOP_WRAP);
            unsigned char code = 0;
            if (done != PARC_DONE_NULL || motion_end) code = (done == PARC_DONE_FAIL) ? 1 : 2;
            if (motion_end) done = PARC_DONE_FAIL;
            if (P.never_done) done = PARC_DONE_NULL; // ig_parkour_env.py:980: after update_done (the curriculum still sees the episode end)
            if (valid && i == 0) {
                P.buf.reward[e_j] = rew;
                P.buf.done[e_j] = done;
                P.ema_code[e_j] = code;
                P.buf.timestep[e_j] = ts_j;
                if (P.buf.time) P.buf.time[e_j] = time_j;
            }
            if (P.buf.reward_terms && valid && i < 7) {
                const float vals[7] = {pose_r, vel_r, root_pose_r, root_vel_r, key_pos_r, contact_pen, rew};
                float v = vals[0];
#pragma unroll
                for (int qq = 1; qq < 7; ++qq) v = i == qq ? vals[qq] : v;
                P.buf.reward_terms[(size_t)i * P.N + e_j] = v;
            }
        }

        // optional outputs: ref body positions / dof positions, tracking error
        if (MIRROR && P.buf.ref_body_pos && lane < B) {
            float *o = P.buf.ref_body_pos + 3 * ((size_t)e * B + lane);
            const float4 v = s_fk[16 + lane];
            o[0] = v.x; o[1] = v.y; o[2] = v.z;
        }
        if (MIRROR && P.buf.ref_dof_pos && lane >= 1 && lane < B) { // kin_char_model.py:601
            const int ty = T->h.jtype[lane];
            float out3[3] = {0.f, 0.f, 0.f};
            joint_rot_to_dof(ty, T->h.axis[lane], s_q[1][lane], out3);
            const int nd = ty == PARC_JOINT_HINGE ? 1 : (ty == PARC_JOINT_SPHERICAL ? 3 : 0);
            for (int k = 0; k < nd; ++k) P.buf.ref_dof_pos[(size_t)e * D + T->h.dof_idx[lane] + k] = out3[k];
        }
        if (MIRROR && P.tracking && P.buf.tracking_error) { // mgdm_dm_util.py:521-553 (per wave: an optional output, off in training)
            const float4 trp = s_q[1][15], tv = s_refvel[0], tav = s_refvel[1];
            const float root_rot_angle = quat_diff_angle(s_q[0][0], s_q[1][0]);
            const float rvx = lane_value(aux0, 30), rvy = lane_value(aux1, 30), rvz = lane_value(aux2, 30);
            const float rax = lane_value(aux0, 31), ray_ = lane_value(aux1, 31), raz = lane_value(aux2, 31);
            float e_rot = 0.f, e_pos = 0.f, e_dv = 0.f;
            if (lane < B) {
                e_rot = fabsf(quat_diff_angle(s_br[0][lane], s_br[1][lane]));
                const float4 cb = s_fk[lane], rb = s_fk[16 + lane];
                e_pos = norm3(mk3((rb.x - trp.x) - (cb.x - root_pos.x), (rb.y - trp.y) - (cb.y - root_pos.y), (rb.z - trp.z) - (cb.z - root_pos.z)));
            }
            if (lane < D) e_dv = fabsf(((const float *)s_refvel)[8 + lane] - s_cdofv[lane]);
            const float pe = wave_sum(e_rot), bpe = wave_sum(e_pos), dve = wave_sum(e_dv);
            if (lane == 0) {
                float *te = P.buf.tracking_error + 7 * (size_t)e;
                te[0] = norm3(mk3(trp.x - root_pos.x, trp.y - root_pos.y, trp.z - root_pos.z));
                te[1] = fabsf(root_rot_angle);
                te[2] = bpe / (float)B;
                te[3] = pe / (float)B;
                te[4] = dve / (float)D;
                te[5] = (fabsf(tv.x - rvx) + fabsf(tv.y - rvy) + fabsf(tv.z - rvz)) / 3.f;
                te[6] = (fabsf(tav.x - rax) + fabsf(tav.y - ray_) + fabsf(tav.z - raz)) / 3.f;
            }
        }
    }
    if (P.body_pos_from_fk && P.buf.char_body_pos && lane < B) {
        float *o = P.buf.char_body_pos + 3 * ((size_t)e * B + lane);
        const float4 v = s_fk[lane];
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
    }
    WAVE_SYNC();
    STAMP(6);

    // ================= stream the staged observation prefix out: 16 bytes per lane per store =================
    {
        const int nst = P.off_tarc; // [0, off_tarc) staged; contacts and rays were stored directly
        if ((P.obs_dim & 3) == 0) {
            const float4 *src = (const float4 *)s_obs;
            float4 *dst = (float4 *)orow;
            const int n4 = nst >> 2;
            for (int i = lane; i < n4; i += 64) dst[i] = src[i];
            for (int i = 4 * n4 + lane; i < nst; i += 64) orow[i] = s_obs[i];
        } else {
            for (int i = lane; i < nst; i += 64) orow[i] = s_obs[i];
        }
    }
    STAMP(7);
#ifdef PARC_STAMPS
    WAVE_SYNC();
    if (lane < 7 && P.stamp_out) P.stamp_out[(size_t)e * 8 + lane] = (unsigned)(s_stamp[lane + 1] - s_stamp[lane]);
    if (lane == 7 && P.stamp_out) P.stamp_out[(size_t)e * 8 + 7] = 0;
#endif
}

// ------------------------------------------------------------------------------------------------
// curriculum: sequential fail-rate EMA in env order (dm_env.py:646-660), exact op order per motion.
//   k_done_scatter: ordered (stable) compaction of the finished envs, one block per 1024 envs; the per-chunk
//                   totals were counted by the step kernel, so a block's base offset is a 128-term sum.
//   k_fail_rate_ema: one wave per motion walks the env-ordered list 64 entries at a time and applies
//                   f <- f*(1-w)+w (FAIL) / f <- f*(1-w) (TIME, SUCC, motion end) in that order.
// The compacted list doubles as the env-id list of parc_env_reset_done (no nonzero(), no host sync).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nonzero_bytes(unsigned v) {
    return ((v & 0xffu) != 0) + ((v & 0xff00u) != 0) + ((v & 0xff0000u) != 0) + ((v & 0xff000000u) != 0);
}

// Block b owns envs [1024 b, 1024 b + 1024).  Its base offset = number of finished envs before it, which it
// counts itself from the (1 byte per env) code array: no atomics in the step kernel, no second launch.
__device__ __forceinline__ void done_scatter_block(const unsigned char *__restrict__ ema_code, const int *__restrict__ motion_ids, int N, int *done_list,
                                                   int *done_key, int *reset_count, int never_done, int b, int nblocks) {
    __shared__ int s_base, s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    {
        const uint4 *src = (const uint4 *)ema_code; // hipMalloc'd: 16-byte aligned; 1024 b bytes = 64 b uint4
        int c = 0;
        for (int i = tid; i < 64 * b; i += 1024) {
            const uint4 v = src[i];
            c += nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
        }
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (lane == 0 && c) atomicAdd(&s_base, c);
    }
    const int e = b * 1024 + tid;
    const unsigned char code = e < N ? ema_code[e] : 0;
    const unsigned long long mask = __ballot(code != 0);
    const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wv] = __popcll(mask);
    __syncthreads();
    int woff = 0, total = 0;
    for (int i = 0; i < 16; ++i) { const int c = s_wave[i]; if (i < wv) woff += c; total += c; }
    if (code) {
        const int pos = s_base + woff + prefix;
        done_list[pos] = e;
        done_key[pos] = (motion_ids[e] << 1) | (code == 1 ? 1 : 0);
    }
    if (b == nblocks - 1 && tid == 0) {
        reset_count[0] = s_base + total;                   // entries of the list: the fail-rate EMA walks all of them
        reset_count[1] = never_done ? 0 : s_base + total;  // envs parc_env_reset_done resets (never_done: flags are NULL, nobody)
    }
}
// The health counters of the dynamics kernel (flag waits that timed out, contact planes that found no slot) go to two words of
// host-mapped pinned memory with the first launch after every step: the host reads them without a synchronisation or a copy
// (HipParkourEnv.step looks at them every step: a protocol error aborts the run at the next step instead of at the next log interval).
__device__ __forceinline__ void publish_health(unsigned int *health) {
    if (health && blockIdx.x == 0 && threadIdx.x == 0) {
        __builtin_nontemporal_store(parcdyn::g_wave_timeouts, health);
        __builtin_nontemporal_store(parcdyn::g_wave_man_drops, health + 1);
    }
}

__global__ __launch_bounds__(1024) void k_done_scatter(const unsigned char *__restrict__ ema_code, const int *__restrict__ motion_ids,
                                                       int N, int *done_list, int *done_key, int *reset_count, int never_done, unsigned int *health) {
    publish_health(health);
    done_scatter_block(ema_code, motion_ids, N, done_list, done_key, reset_count, never_done, blockIdx.x, gridDim.x);
}

// One 1024-thread block per motion.  The block sweeps the env-ordered list and compacts (stable, by rank) the addends of its
// motion's entries -- w for FAIL, +0 otherwise -- into LDS; thread 0 then applies the chain in that order:
// f <- f*(1-w) + addend  (adding +0.0f leaves f*(1-w) unchanged, so this is the reference's two-branch update,
// dm_env.py:651-658, bit for bit).  The chain is inherently serial (each update rounds): only its two dependent VALU
// operations per entry stay on it, the addends arrive four at a time from LDS.
#define EMA_CAP 8192
// the chain itself, on one thread: f <- f*keep + addend over `n` buffered addends (eight LDS reads in flight: only the first one's latency is exposed)
__device__ __forceinline__ float ema_chain(float f, const float *s_add, int n, float keep) {
    int i4 = 0;
    const float4 *a4 = (const float4 *)s_add;
    for (; i4 + 32 <= n; i4 += 32) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = a4[(i4 >> 2) + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f = f * keep; f = f + q[j].x;
            f = f * keep; f = f + q[j].y;
            f = f * keep; f = f + q[j].z;
            f = f * keep; f = f + q[j].w;
        }
    }
    for (; i4 + 4 <= n; i4 += 4) {
        const float4 a = a4[i4 >> 2];
        f = f * keep; f = f + a.x;
        f = f * keep; f = f + a.y;
        f = f * keep; f = f + a.z;
        f = f * keep; f = f + a.w;
    }
    for (; i4 < n; ++i4) { f = f * keep; f = f + s_add[i4]; }
    return f;
}
// One pass per EMA_CAP entries of the env-ordered key list ((motion << 1) | fail bit): every thread takes 8 consecutive entries (two 16-byte
// loads), a block-wide exclusive scan of the per-thread match counts places the motion's addends in env order, thread 0 runs the chain.
// (Round 3: the list was swept 1 024 entries at a time with two barriers per sweep -- 1.5 us per sweep, six sweeps per step at 65 536 envs.)
__device__ __forceinline__ void fail_rate_ema_block(const int *__restrict__ done_key, int k, float *fail_rates, int m, float w) {
    __shared__ __align__(16) float s_add[EMA_CAP];
    __shared__ int s_wtot[16];
    __shared__ float s_f;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float keep = (float)(1.0 - (double)w);
    if (tid == 0) s_f = fail_rates[m];
    bool any = false;
    for (int base = 0; base < k; base += EMA_CAP) {
        const int e0 = base + 8 * tid;
        int key[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) key[j] = -2;
        if (e0 + 8 <= k) { // the list buffer is 16-byte aligned and e0 a multiple of 8
            const int4 a = *(const int4 *)(done_key + e0), c = *(const int4 *)(done_key + e0 + 4);
            key[0] = a.x; key[1] = a.y; key[2] = a.z; key[3] = a.w; key[4] = c.x; key[5] = c.y; key[6] = c.z; key[7] = c.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (e0 + j < k) key[j] = done_key[e0 + j];
        }
        unsigned matchbits = 0u, failbits = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (key[j] >= 0 && (key[j] >> 1) == m) { matchbits |= 1u << j; if (key[j] & 1) failbits |= 1u << j; }
        const int cnt = __popc(matchbits);
        int incl = cnt; // inclusive scan within the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
        __syncthreads(); // s_f / the previous pass's buffer are settled
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; ++q) { const int c = s_wtot[q]; if (q < wv) woff += c; total += c; }
        int pos = woff + incl - cnt;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (matchbits & (1u << j)) s_add[pos++] = (failbits & (1u << j)) ? w : 0.0f;
        __syncthreads();
        if (total > 0) { // uniform
            any = true;
            if (tid == 0) s_f = ema_chain(s_f, s_add, total, keep);
        }
    }
    __syncthreads();
    if (tid == 0 && any) fail_rates[m] = s_f;
}
// The same chain straight from the step kernel's per-env codes (1 byte per env: 0 = not finished, 1 = FAIL, 2 = other) for shards of up to
// CURRICULUM_ONE_LAUNCH_MAX envs: every thread takes 8 consecutive envs (one 8-byte load; finished envs are few, so their motion ids are
// conditional loads), a block-wide exclusive scan of the per-thread match counts gives each its place in env order, 8 192 envs per pass.
__device__ __forceinline__ void fail_rate_ema_raw_block(const unsigned char *__restrict__ ema_code, const int *__restrict__ motion_ids, int N,
                                                        float *fail_rates, int m, float w) {
    __shared__ __align__(16) float s_add[EMA_CAP];
    __shared__ int s_wtot[16];
    __shared__ float s_f;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float keep = (float)(1.0 - (double)w);
    if (tid == 0) s_f = fail_rates[m];
    bool any = false;
    for (int base = 0; base < N; base += EMA_CAP) { // N is a multiple of 8 on this path (launch_curriculum)
        const int e0 = base + 8 * tid;
        unsigned long long codes = 0ull;
        if (e0 < N) codes = *(const unsigned long long *)(ema_code + e0);
        unsigned matchbits = 0u, failbits = 0u;
        if (codes) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned c = (unsigned)(codes >> (8 * j)) & 0xffu;
                if (c && motion_ids[e0 + j] == m) { matchbits |= 1u << j; if (c == 1u) failbits |= 1u << j; }
            }
        }
        const int cnt = __popc(matchbits);
        int incl = cnt; // inclusive scan within the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
        __syncthreads(); // s_f / the previous pass's buffer are settled
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int woff = 0, total = 0;
        for (int q = 0; q < 16; ++q) { const int c = s_wtot[q]; if (q < wv) woff += c; total += c; }
        int pos = woff + incl - cnt;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (matchbits & (1u << j)) s_add[pos++] = (failbits & (1u << j)) ? w : 0.0f;
        __syncthreads();
        if (total > 0) { // uniform
            any = true;
            if (tid == 0) s_f = ema_chain(s_f, s_add, total, keep);
        }
    }
    __syncthreads();
    if (tid == 0 && any) fail_rates[m] = s_f;
}
__global__ __launch_bounds__(1024) void k_fail_rate_ema(const int *__restrict__ done_key, const int *__restrict__ reset_count,
                                                       float *fail_rates, int M, float w) {
    const int k = *reset_count, m = blockIdx.x;
    if (k == 0 || m >= M) return;
    fail_rate_ema_block(done_key, k, fail_rates, m, w);
}
// Both of the above as ONE launch, for shards of up to CURRICULUM_ONE_LAUNCH_MAX envs (what a GPU of an 8-GPU job runs): blocks [0, nchunks) compact the finished envs
// (the reset list), blocks nchunks.. apply the EMA of motion m -- reading the step kernel's per-env codes directly instead of the compacted
// list, so that no block waits for another.  Same entries in the same (env) order: bit-identical to the two launches.
#define CURRICULUM_ONE_LAUNCH_MAX 8192 // measured: -1.6 us per step at 8 192 envs, +1.6 us at 16 384 (two passes of the scan)
__global__ __launch_bounds__(1024) void k_curriculum_small(const unsigned char *__restrict__ ema_code, const int *__restrict__ motion_ids, int N,
                                                          int *done_list, int *done_key, int *reset_count, int never_done, int nchunks,
                                                          float *fail_rates, int M, float w, unsigned int *health) {
    publish_health(health);
    if ((int)blockIdx.x < nchunks) {
        done_scatter_block(ema_code, motion_ids, N, done_list, done_key, reset_count, never_done, blockIdx.x, nchunks);
    } else {
        const int m = (int)blockIdx.x - nchunks;
        if (m >= M) return;
        fail_rate_ema_raw_block(ema_code, motion_ids, N, fail_rates, m, w);
    }
}

// Large libraries (thousands of motions, a handful of finished envs each): one block per motion sweeping the whole list is
// O(M k).  Instead the FIRST list entry of every motion that occurs becomes its leader (atomicMin over a per-motion slot
// that is kept at INT_MAX between steps), and the leader walks the list once, applying its motion's entries in list
// (= env) order: the same chain, the same roundings, O(k) per motion that actually finished an env.
__global__ void k_ema_first(const int *__restrict__ done_key, const int *__restrict__ reset_count, int *first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *reset_count) return;
    atomicMin(&first[done_key[i] >> 1], i);
}

__global__ __launch_bounds__(256) void k_ema_leader(const int *__restrict__ done_key, const int *__restrict__ reset_count, int *first,
                                                    float *fail_rates, float w) {
    // one WAVE per list entry; only the waves of leaders do anything.  The 64 lanes sweep the rest of the list 64 entries at
    // a time (one coalesced load, two ballots); lane 0 applies the matching entries of a tile in list order.
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int k = *reset_count;
    if (i >= k) return;
    const int m = done_key[i] >> 1;
    if (first[m] != i) return;
    const float keep = (float)(1.0 - (double)w);
    float f = fail_rates[m];
    for (int j0 = i & ~63; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        const int key = j < k ? done_key[j] : -2;
        const bool match = j >= i && (key >> 1) == m;
        unsigned long long mm = __ballot(match);
        const unsigned long long ff = __ballot(match && (key & 1));
        while (mm) { // uniform: every lane runs the same chain, lane 0 stores
            const int t = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            f = f * keep;
            f = f + (((ff >> t) & 1ull) ? w : 0.0f);
        }
    }
    if (lane == 0) { fail_rates[m] = f; first[m] = 0x7fffffff; }
}

// ------------------------------------------------------------------------------------------------
// motion library preparation (motion_lib.py:305-327, kin_char_model.py:651-693): one thread per frame
// ------------------------------------------------------------------------------------------------
struct PrepParams {
    int F, B, J, D;
    const float *root_pos, *root_rot, *joint_rot, *contacts;
    const int *frame_motion; // [F]
    const MotionMeta *meta;
    const DevTables *tables;
    float4 *records;
};

__global__ void k_motion_prep(const PrepParams P) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= P.F) return;
    const int m = P.frame_motion[f];
    const MotionMeta meta = P.meta[m];
    const DevTables *T = P.tables;
    float4 *rec = P.records + (size_t)f * REC_F4;
    float *recf = (float *)rec;
    for (int i = 0; i < 128; ++i) recf[i] = 0.f;
    const float *rr = P.root_rot + 4 * (size_t)f;
    rec[0] = make_float4(rr[0], rr[1], rr[2], rr[3]);
    for (int j = 0; j < P.J; ++j) {
        const float *q = P.joint_rot + 4 * ((size_t)f * P.J + j);
        rec[1 + j] = make_float4(q[0], q[1], q[2], q[3]);
    }
    const float *rp = P.root_pos + 3 * (size_t)f;
    rec[REC_Q_POS] = make_float4(rp[0], rp[1], rp[2], 0.f);
    for (int b = 0; b < P.B; ++b) recf[4 * REC_Q_CONTACT + b] = P.contacts ? P.contacts[(size_t)f * P.B + b] : 0.f;
    // velocities: finite difference (f0, f0+1); the last frame repeats the previous pair
    const int local = f - meta.start;
    int f0 = f;
    if (local == meta.nframes - 1) f0 = f - 1;
    if (meta.nframes < 2) return;
    const float fps = meta.fps;
    const float dt = (float)(1.0 / (double)meta.fps);
    const float *p0 = P.root_pos + 3 * (size_t)f0, *p1 = p0 + 3;
    recf[4 * REC_Q_VEL + 0] = fps * (p1[0] - p0[0]);
    recf[4 * REC_Q_VEL + 1] = fps * (p1[1] - p0[1]);
    recf[4 * REC_Q_VEL + 2] = fps * (p1[2] - p0[2]);
    const float *r0 = P.root_rot + 4 * (size_t)f0, *r1 = r0 + 4;
    const Q4 dq = quat_mul(mk4(r1[0], r1[1], r1[2], r1[3]), quat_conj(mk4(r0[0], r0[1], r0[2], r0[3]))); // quat_diff :454
    const V3 ev = quat_to_exp_map(dq);
    recf[4 * (REC_Q_VEL + 1) + 0] = fps * ev.x;
    recf[4 * (REC_Q_VEL + 1) + 1] = fps * ev.y;
    recf[4 * (REC_Q_VEL + 1) + 2] = fps * ev.z;
    float *dv = recf + 4 * (REC_Q_VEL + 2);
    for (int j = 1; j < P.B; ++j) { // compute_dof_vel kin_char_model.py:661
        const float *a = P.joint_rot + 4 * ((size_t)f0 * P.J + j - 1), *b = a + 4 * P.J;
        const Q4 d = quat_normalize(quat_mul(quat_conj(mk4(a[0], a[1], a[2], a[3])), mk4(b[0], b[1], b[2], b[3])));
        const int ty = T->h.jtype[j];
        if (ty == PARC_JOINT_HINGE) {
            V3 x = quat_to_exp_map(d);
            x = mk3(x.x / dt, x.y / dt, x.z / dt);
            dv[T->h.dof_idx[j]] = T->h.axis[j][0] * x.x + T->h.axis[j][1] * x.y + T->h.axis[j][2] * x.z;
        } else if (ty == PARC_JOINT_SPHERICAL) {
            const V3 x = quat_to_exp_map(d);
            dv[T->h.dof_idx[j] + 0] = x.x / dt;
            dv[T->h.dof_idx[j] + 1] = x.y / dt;
            dv[T->h.dof_idx[j] + 2] = x.z / dt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// one-thread-per-item operators (cfg-1 plumbing, reset): calc_motion_frame, dof<->rot, FK
// ------------------------------------------------------------------------------------------------
struct FrameOut { float *root_pos, *root_rot, *root_vel, *root_ang_vel, *joint_rot, *dof_vel, *contacts; };

__device__ void motion_frame_thread(const float4 *records, const MotionMeta meta, float t, int B, int J, int D, size_t o,
                                    const FrameOut &F) {
    const Blend bl = frame_blend(meta, t);
    const float4 *r0 = records + (size_t)bl.i0 * REC_F4, *r1 = records + (size_t)bl.i1 * REC_F4;
    const float b = bl.b, a = 1.0f - b;
    {
        const float4 A = r0[REC_Q_POS], Bv = r1[REC_Q_POS];
        float x = a * A.x + b * Bv.x, y = a * A.y + b * Bv.y, z = a * A.z + b * Bv.z;
        if (meta.loop == PARC_LOOP_WRAP) {
            const float ph = floorf(t / meta.length);
            x = x + ph * meta.dx; y = y + ph * meta.dy; z = z + ph * meta.dz;
        }
        F.root_pos[3 * o] = x; F.root_pos[3 * o + 1] = y; F.root_pos[3 * o + 2] = z;
    }
    const Q4 rr = slerp(r0[0], r1[0], b);
    *(float4 *)(F.root_rot + 4 * o) = rr;
    for (int j = 0; j < J; ++j) *(float4 *)(F.joint_rot + 4 * (o * J + j)) = slerp(r0[1 + j], r1[1 + j], b);
    const float *v = (const float *)(r0 + REC_Q_VEL);
    if (F.root_vel) { F.root_vel[3 * o] = v[0]; F.root_vel[3 * o + 1] = v[1]; F.root_vel[3 * o + 2] = v[2]; }
    if (F.root_ang_vel) { F.root_ang_vel[3 * o] = v[4]; F.root_ang_vel[3 * o + 1] = v[5]; F.root_ang_vel[3 * o + 2] = v[6]; }
    if (F.dof_vel) for (int d = 0; d < D; ++d) F.dof_vel[o * D + d] = v[8 + d];
    if (F.contacts) {
        const float *c0 = (const float *)(r0 + REC_Q_CONTACT), *c1 = (const float *)(r1 + REC_Q_CONTACT);
        for (int k = 0; k < B; ++k) F.contacts[o * B + k] = a * c0[k] + b * c1[k];
    }
}

__global__ void k_calc_motion_frame(const float4 *records, const MotionMeta *meta, const int *ids, const float *times, int n,
                                    int B, int J, int D, FrameOut F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    motion_frame_thread(records, meta[ids[i]], times[i], B, J, D, (size_t)i, F);
}

__global__ void k_dof_to_rot(const DevTables *T, const float *dof, float *jr, int n, int B, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 1; j < B; ++j)
        *(float4 *)(jr + 4 * ((size_t)i * (B - 1) + j - 1)) = joint_dof_to_rot(T->h.jtype[j], T->h.axis[j], dof + (size_t)i * D + T->h.dof_idx[j]);
}

__global__ void k_rot_to_dof(const DevTables *T, const float *jr, float *dof, int n, int B, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int d = 0; d < D; ++d) dof[(size_t)i * D + d] = 0.f;
    for (int j = 1; j < B; ++j)
        joint_rot_to_dof(T->h.jtype[j], T->h.axis[j], *(const float4 *)(jr + 4 * ((size_t)i * (B - 1) + j - 1)), dof + (size_t)i * D + T->h.dof_idx[j]);
}

__device__ void fk_thread(const DevTables *T, int B, V3 root_pos, Q4 root_rot, const float *jr, float *bp, float *br) {
    Q4 rot[PARC_MAX_BODIES];
    V3 pos[PARC_MAX_BODIES];
    rot[0] = root_rot; pos[0] = root_pos;
    for (int j = 1; j < B; ++j) {
        const int p = T->h.parent[j];
        const V3 wt = quat_rotate(rot[p], mk3(T->h.lt[j][0], T->h.lt[j][1], T->h.lt[j][2]));
        pos[j] = mk3(pos[p].x + wt.x, pos[p].y + wt.y, pos[p].z + wt.z);
        const Q4 q = *(const float4 *)(jr + 4 * (j - 1));
        rot[j] = quat_mul(rot[p], quat_mul(mk4(T->h.lr[j][0], T->h.lr[j][1], T->h.lr[j][2], T->h.lr[j][3]), q));
    }
    for (int j = 0; j < B; ++j) {
        bp[3 * j] = pos[j].x; bp[3 * j + 1] = pos[j].y; bp[3 * j + 2] = pos[j].z;
        if (br) *(float4 *)(br + 4 * j) = rot[j];
    }
}

__global__ void k_fk(const DevTables *T, const float *root_pos, const float *root_rot, const float *jr, float *bp, float *br,
                     int n, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *rp = root_pos + 3 * (size_t)i, *rr = root_rot + 4 * (size_t)i;
    fk_thread(T, B, mk3(rp[0], rp[1], rp[2]), mk4(rr[0], rr[1], rr[2], rr[3]), jr + 4 * (size_t)i * (B - 1),
              bp + 3 * (size_t)i * B, br ? br + 4 * (size_t)i * B : nullptr);
}

// ---- device RNG for resets: Philox4x32-10 ---------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned long long ctr_hi, unsigned ctr_lo, float *u4) {
    unsigned c0 = ctr_lo, c1 = (unsigned)ctr_hi, c2 = (unsigned)(ctr_hi >> 32), c3 = 0x5041524Bu;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    u4[0] = (c0 >> 8) * (1.0f / 16777216.0f); u4[1] = (c1 >> 8) * (1.0f / 16777216.0f);
    u4[2] = (c2 >> 8) * (1.0f / 16777216.0f); u4[3] = (c3 >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_up((int)(b & 0xffffffffll), d, 64), hi = __shfl_up((int)(b >> 32), d, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}


// ---- sampling of the reset state (dm_env.py:470-504), inside k_reset_with --------------------------------------------
// The Philox call index of a sampling reset is *call_dev + 1; the observation launch that follows the reset kernel bumps
// the device-side counter (so that a captured graph replays with a fresh index every time).
#define CDF_LOCAL_MAX 64
struct SampleParams {
    int enabled;                 // 0: the caller passes motion ids / terrain ids / start times / xy noise (parc_env_reset_with)
    int M, T, rand_reset, demo_mode;
    float min_w, noise_scale;
    const float *cdf_global;     // k_build_cdf's output, libraries of more than CDF_LOCAL_MAX motions
    const float *fail_rates, *motion_weights, *start_frac;
    unsigned long long seed;
    const unsigned long long *call_dev;
};

__device__ __forceinline__ void sample_reset(const SampleParams &SP, const float *cdf, const MotionMeta *meta, int e, int &mid, int &tid, float &t0,
                                             float &nx, float &ny) {
    const unsigned long long call = *SP.call_dev + 1ull;
    float u[4], v[4];
    philox4(SP.seed, call, (unsigned)e * 2u, u);
    philox4(SP.seed, call, (unsigned)e * 2u + 1u, v);
    const int M = SP.M;
    if (SP.demo_mode) mid = e % M; // dm_env.py:479-480
    else { // multinomial with replacement == inverse CDF per draw (motion_lib.py:56-60)
        const float x = u[0] * cdf[M - 1];
        int lo = 0, hi = M - 1;
        while (lo < hi) { const int md = (lo + hi) >> 1; if (cdf[md] > x) hi = md; else lo = md + 1; }
        mid = lo;
    }
    tid = min((int)(u[1] * (float)SP.T), SP.T - 1);                                       // torch.randint(high=T)
    const float len = meta[mid].length;
    t0 = SP.rand_reset ? u[2] * len : len * (SP.start_frac ? SP.start_frac[e] : 0.f);     // motion_lib.py:62-72, dm_env.py:500-504
    nx = SP.noise_scale * (v[0] * 2.0f - 1.0f);                                           // mgdm_dm_util.py:103-104
    ny = SP.noise_scale * (v[1] * 2.0f - 1.0f);
}

// ------------------------------------------------------------------------------------------------
// reset (dm_env.py:567-592): thread per reset env
// ------------------------------------------------------------------------------------------------
struct ResetParams {
    int B, J, D, T, N;
    const float4 *records; const MotionMeta *meta; const float *motion_offsets; const float *env_offsets;
    const DevTables *tables;
    float *scratch_jr;  // [N][J][4] when ref_joint_rot is not bound
    float4 *prep;       // [N][16]: the record k_env_prep would write for the reset envs (saves that launch)
    ParcEnvBuffers buf;
};

// 16 lanes per env (4 envs per wave): lane j owns quaternion / body j of the character, lane 15 the root position.  Same
// arithmetic per quantity as the one-thread-per-env form it replaces (motion_frame_thread, joint_rot_to_dof, fk_thread),
// so results are unchanged; the per-env serial chain (15 slerps + 14 exp maps + 15-body FK, ~7k instructions) becomes ~600.
__global__ __launch_bounds__(64) void k_reset_with(const ResetParams P, const int64_t *env_ids, const int *env_ids32, const int *count_dev, int k,
                                                   const int *motion_ids, const int *terrain_ids, const float *t0, const float *xy_noise,
                                                   const SampleParams SP) {
    __shared__ float s_cdf[CDF_LOCAL_MAX];
    __shared__ float4 s_q[4][16];     // [g][0] root rotation, [g][j] joint j (j >= 1)
    __shared__ float4 s_pos[4][16];   // FK: body positions
    __shared__ float4 s_rot[4][16];   // FK: body rotations
    __shared__ float s_dof[4][PARC_MAX_DOFS];
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int i = blockIdx.x * 4 + g;
    if (count_dev) k = *count_dev;
    if ((int)blockIdx.x * 4 >= k) return; // the grid is sized for all envs, the list is usually short
    const bool live = i < k;
    const int B = P.B, J = P.J, D = P.D;
    int e = 0, mid = 0, tid = 0;
    float t = 0.f;
    Blend bl; bl.i0 = 0; bl.i1 = 0; bl.b = 0.f;
    MotionMeta meta = P.meta[0];
    float nx = 0.f, ny = 0.f; // add_noise_to_char_state (mgdm_dm_util.py:102-106): xy += scale * U(-1,1)
    if (live) e = env_ids ? (int)env_ids[i] : (env_ids32 ? env_ids32[i] : i);
    if (SP.enabled) {
        const float *cdf = SP.cdf_global;
        if (SP.M <= CDF_LOCAL_MAX) { // the wave scan of k_build_cdf for a library that fits one wave: same operations, same result
            const int lane = threadIdx.x;
            double v = lane < SP.M ? (double)(fmaxf(SP.fail_rates[lane], SP.min_w) * SP.motion_weights[lane]) : 0.0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const double u = shfl_up_f64(v, d); if (lane >= d) v += u; }
            if (lane < SP.M) s_cdf[lane] = (float)(0.0 + v);
            __syncthreads();
            cdf = s_cdf;
        }
        if (live && j == 0) sample_reset(SP, cdf, P.meta, e, mid, tid, t, nx, ny); // lane 0 of the env's 16 draws, the others receive
        mid = __shfl(mid, g * 16, 64); tid = __shfl(tid, g * 16, 64); t = __shfl(t, g * 16, 64);
        nx = __shfl(nx, g * 16, 64); ny = __shfl(ny, g * 16, 64);
    } else if (live) {
        mid = motion_ids[i]; tid = terrain_ids[i]; t = t0[i];
        nx = xy_noise[2 * i]; ny = xy_noise[2 * i + 1];
    }
    if (live) {
        meta = P.meta[mid];
        bl = frame_blend(meta, t);
    }
    const float4 *r0 = P.records + (size_t)bl.i0 * REC_F4, *r1 = P.records + (size_t)bl.i1 * REC_F4;
    const float b = bl.b, a = 1.0f - b;
    float *jr_out = P.buf.ref_joint_rot ? P.buf.ref_joint_rot : P.scratch_jr;
    for (int d = j; d < D; d += 16) s_dof[g][d] = 0.f;
    // ---- reference frame at t0 -> character state (mgdm_dm_util.py:89-100); ref_* mirrors when bound
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (live) {
        if (j < B) {
            const Q4 q = slerp(r0[j], r1[j], b);
            s_q[g][j] = q;
            if (j == 0) {
                *(float4 *)(P.buf.char_root_rot + 4 * (size_t)e) = q;
                if (P.buf.ref_root_rot) *(float4 *)(P.buf.ref_root_rot + 4 * (size_t)e) = q;
            } else {
                *(float4 *)(jr_out + 4 * ((size_t)e * J + j - 1)) = q;
            }
        }
        // every lane forms the root position (the FK lanes need it too); lane 15 stores it
        const float4 A = r0[REC_Q_POS], Bv = r1[REC_Q_POS];
        float x = a * A.x + b * Bv.x, y = a * A.y + b * Bv.y, z = a * A.z + b * Bv.z;
        if (meta.loop == PARC_LOOP_WRAP) {
            const float ph = floorf(t / meta.length);
            x = x + ph * meta.dx; y = y + ph * meta.dy; z = z + ph * meta.dz;
        }
        const float *mo = P.motion_offsets + 2 * ((size_t)mid * P.T + tid);
        const float ox = mo[0] - P.env_offsets[3 * e], oy = mo[1] - P.env_offsets[3 * e + 1];
        rx = x + ox; ry = y + oy; rz = z;
        // velocities of frame idx0 (un-interpolated) and contacts, spread over the lanes
        const float *v = (const float *)(r0 + REC_Q_VEL);
        if (j < 3) {
            P.buf.char_root_vel[3 * (size_t)e + j] = v[j];
            P.buf.char_root_ang_vel[3 * (size_t)e + j] = v[4 + j];
            if (P.buf.ref_root_vel) P.buf.ref_root_vel[3 * (size_t)e + j] = v[j];
            if (P.buf.ref_root_ang_vel) P.buf.ref_root_ang_vel[3 * (size_t)e + j] = v[4 + j];
        }
        for (int d = j; d < D; d += 16) {
            P.buf.char_dof_vel[(size_t)e * D + d] = v[8 + d];
            if (P.buf.ref_dof_vel) P.buf.ref_dof_vel[(size_t)e * D + d] = v[8 + d];
        }
        if (P.buf.ref_contacts && j < B) {
            const float *c0 = (const float *)(r0 + REC_Q_CONTACT), *c1 = (const float *)(r1 + REC_Q_CONTACT);
            P.buf.ref_contacts[(size_t)e * B + j] = a * c0[j] + b * c1[j];
        }
    }
    __syncthreads();
    if (live && j >= 1 && j < B) joint_rot_to_dof(P.tables->h.jtype[j], P.tables->h.axis[j], s_q[g][j], &s_dof[g][P.tables->h.dof_idx[j]]);
    __syncthreads();
    // the record k_env_prep forms from the NEW state (same expressions: heading terms from the root rotation, character
    // joint quaternions from the stored dofs), so the observation pass that follows needs no prep launch
    if (live && j < B) {
        float4 *out = P.prep + (size_t)e * 16;
        if (j == 0) {
            const float heading = calc_heading(s_q[g][0]);
            const Q4 hinv = heading_quat_inv(heading);
            out[0] = make_float4(cosf(heading), sinf(heading), hinv.z, hinv.w);
        } else {
            const int ty = P.tables->h.jtype[j], di = P.tables->h.dof_idx[j];
            Q4 q = mk4(0.f, 0.f, 0.f, 1.f);
            if (ty == PARC_JOINT_HINGE) q = axis_angle_to_quat(mk3(P.tables->h.axis[j][0], P.tables->h.axis[j][1], P.tables->h.axis[j][2]), s_dof[g][di]);
            else if (ty == PARC_JOINT_SPHERICAL) q = exp_map_to_quat(mk3(s_dof[g][di], s_dof[g][di + 1], s_dof[g][di + 2]));
            out[j] = q;
        }
    }
    float cx = rx, cy = ry;
    if (live) { cx = rx + nx; cy = ry + ny; }
    if (live) {
        for (int d = j; d < D; d += 16) {
            P.buf.char_dof_pos[(size_t)e * D + d] = s_dof[g][d];
            if (P.buf.ref_dof_pos) P.buf.ref_dof_pos[(size_t)e * D + d] = s_dof[g][d];
        }
        if (j == 15) {
            if (P.buf.ref_root_pos) { float *o = P.buf.ref_root_pos + 3 * (size_t)e; o[0] = rx; o[1] = ry; o[2] = rz; }
            float *crp = P.buf.char_root_pos + 3 * (size_t)e;
            crp[0] = cx; crp[1] = cy; crp[2] = rz;
            P.buf.motion_ids[e] = mid;
            P.buf.terrain_ids[e] = tid;
            P.buf.time_offsets[e] = t;
            P.buf.timestep[e] = 0;
            if (P.buf.time) P.buf.time[e] = 0.f;
            P.buf.done[e] = PARC_DONE_NULL;
            if (P.buf.ep_num) P.buf.ep_num[e] += 1; // ig_parkour_env.py:826-827
        }
        if (P.buf.contact_forces) for (int c = j; c < 3 * B; c += 16) P.buf.contact_forces[(size_t)e * 3 * B + c] = 0.f;
    }
    // rigid bodies: FK of the new state (PhysX would refresh them one sim step later).  Lane j walks root -> body j; each
    // body's pose is formed from its parent's exactly as fk_thread does, level by level.
    if (P.buf.char_body_pos) {
        if (live && j == 0) { s_rot[g][0] = s_q[g][0]; s_pos[g][0] = make_float4(cx, cy, rz, 0.f); }
        __syncthreads();
        int dj = 0; // depth of body j in the tree
        if (j >= 1 && j < B) for (int q = j; q != 0; q = P.tables->h.parent[q]) ++dj;
        for (int depth = 1; depth < B; ++depth) { // level by level: body j is formed once its parent is
            if (live && j >= 1 && j < B) {
                if (dj == depth) {
                    const int p = P.tables->h.parent[j];
                    const Q4 rp_ = s_rot[g][p];
                    const float4 pp = s_pos[g][p];
                    const V3 wt = quat_rotate(rp_, mk3(P.tables->h.lt[j][0], P.tables->h.lt[j][1], P.tables->h.lt[j][2]));
                    s_pos[g][j] = make_float4(pp.x + wt.x, pp.y + wt.y, pp.z + wt.z, 0.f);
                    s_rot[g][j] = quat_mul(rp_, quat_mul(mk4(P.tables->h.lr[j][0], P.tables->h.lr[j][1], P.tables->h.lr[j][2], P.tables->h.lr[j][3]), s_q[g][j]));
                }
            }
            __syncthreads();
            if (depth >= PARC_MAX_FK_DEPTH + 1) break;
        }
        if (live && j < B) {
            float *bp = P.buf.char_body_pos + 3 * ((size_t)e * B + j);
            const float4 pj = s_pos[g][j];
            bp[0] = pj.x; bp[1] = pj.y; bp[2] = pj.z;
        }
    }
}

// weights = clamp(fail_rate, min_w) * motion_weight (dm_env.py:487-490) -> inclusive CDF (one block)
// Libraries of more than CDF_LOCAL_MAX motions; smaller ones are scanned by k_reset_with itself (one launch less on the reset path).
__global__ __launch_bounds__(1024) void k_build_cdf(const float *fail_rates, const float *motion_weights, float min_w, int M, float *cdf) {
    // rows of 1024 consecutive motions (coalesced); inside a row: wave scans through shuffles, the 16 wave totals through LDS;
    // the running total carries from row to row.  Sums are formed in double, the stored value is its fp32 rounding.
    __shared__ double s_wave[16];
    __shared__ double s_carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0.0;
    __syncthreads();
    for (int base0 = 0; base0 < M; base0 += 16 * 1024) { // 16 rows per pass: their loads are all in flight before the first scan
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = base0 + r * 1024 + threadIdx.x;
            pv[r] = m < M ? fmaxf(fail_rates[m], min_w) * motion_weights[m] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int base = base0 + r * 1024;
            if (base >= M) break; // uniform
            const int m = base + threadIdx.x;
            double v = (double)pv[r];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const double u = shfl_up_f64(v, d); if (lane >= d) v += u; }
            if (lane == 63) s_wave[wv] = v;
            __syncthreads();
            double off = s_carry;
            for (int q = 0; q < wv; ++q) off += s_wave[q];
            if (m < M) cdf[m] = (float)(off + v);
            __syncthreads();
            if (threadIdx.x == 1023) s_carry = off + v;
            __syncthreads();
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(PARC_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

// ParcEnvConfig::dev_options, "key=value;key=value"
static std::string dev_opt(const ParcEnvConfig *cfg, const char *key) {
    if (!cfg->dev_options) return "";
    const std::string all = cfg->dev_options, k = std::string(key) + "=";
    size_t at = 0;
    while (at < all.size()) {
        size_t end = all.find(';', at);
        if (end == std::string::npos) end = all.size();
        if (all.compare(at, k.size(), k) == 0) return all.substr(at + k.size(), end - at - k.size());
        at = end + 1;
    }
    return "";
}

// k_dynamics_wave, envs per block.  A block's latency does not depend on how many of its 64 lanes carry an env, so when 64-env blocks
// would leave half the CUs without one (a multi-GPU shard), 32-env blocks put the same work on twice the CUs: -5 % at 8 192 envs (less LDS
// and memory traffic per block; the vector instructions themselves take as long with 32 lanes as with 64)
static int wave_envs_per_block(int N, int num_cus) { return (N + 63) / 64 <= num_cus / 2 ? 32 : 64; }

struct ParcEnv {
    ParcEnvConfig cfg;
    int B, J, D, K, S, R, N, M = 0, T = 1;
    int obs_dim;
    int64_t F = 0;
    bool bound = false, have_motions = false, have_terrain = false;
    bool done_list_fresh = false;     // a step has produced a done list that parc_env_reset_done has not consumed yet
    StepParams sp;
    float4 *d_prep = nullptr;
    std::string dev_options_s, describe_s; // the dev_options this handle was created with (owned copy) / parc_env_describe
    unsigned int *h_health = nullptr, *d_health = nullptr; // two words of host-mapped pinned memory (publish_health) and their device alias
    float *d_man_ovf = nullptr;       // k_dynamics_wave: overflow area of the per-lane contact-plane lists, [blocks][4 waves][WV_MAN_OVF][8][64]
    float *d_root_shadow = nullptr;   // [N][6]: root position the dynamics last wrote + what that write rounded away (k_dynamics_wave)
    parcdyn::DynModel h_dyn;
    parcdyn::DynModel *d_dyn = nullptr;
    parcdyn::CoopTables h_coop;
    parcdyn::CoopTables *d_coop = nullptr;
    bool use_coop = false;
    parcdyn::WaveTables h_wave;
    parcdyn::WaveTables *d_wave = nullptr;
    bool use_wave = false, wave_rejected = false;
    DevTables h_tab;
    DevTables *d_tab = nullptr;
    float *d_ray = nullptr, *d_env_off = nullptr, *d_hf = nullptr, *d_motion_off = nullptr;
    float4 *d_records = nullptr;
    MotionMeta *d_meta = nullptr;
    std::vector<MotionMeta> h_meta;
    std::vector<float> h_weights;
    float *d_weights = nullptr, *d_fail = nullptr, *d_cdf = nullptr;
    unsigned char *d_ema = nullptr;
    int *d_done_list = nullptr, *d_done_key = nullptr, *d_chunk_count = nullptr, *d_motion_done = nullptr, *d_reset_count = nullptr;
    int nchunks = 0;
    float *d_scratch_jr = nullptr, *d_start_frac = nullptr;
    unsigned long long *d_reset_calls = nullptr;   // device counter of sampling resets (Philox call index)
    const float *action_bound = nullptr;           // parc_env_bind_action
    hipGraphExec_t graph_exec = nullptr;           // parc_env_step_reset_graph
    bool graph_dirty = true;
    bool force_ema_leader = false;                 // test switch PARC_EMA_LEADER=1: the large-library EMA path on a small library
    bool force_two_launch_curriculum = false;      // test switch PARC_CURRICULUM_TWO_LAUNCHES=1: k_done_scatter + k_fail_rate_ema on a small shard
    int grid_waves = 0;
    int num_cus = 256;
    size_t lds_bytes = 0;
    float last_dyn_ms = 0.f;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float *rec_frames = nullptr, *rec_obs = nullptr;   // recorder ring buffers (caller-owned)
    int rec_cap = 0, rec_ref = 0;
    int *rec_count = nullptr, *rec_nwriting = nullptr;
    unsigned char *rec_writing = nullptr;
    bool timing = false;               // parc_env_set_kernel_timing: record events around the kernels of every step
    std::vector<hipEvent_t> tev;       // 3 events per timed step (before dynamics, after dynamics, after the obs kernels)
    size_t tev_used = 0;
};

extern "C" const char *parc_last_error(void) { return g_err.c_str(); }
extern "C" int parc_abi_version(void) { return PARC_ABI_VERSION; }

static void free_dev(ParcEnv *e) {
    void *ptrs[] = {e->d_man_ovf, e->d_root_shadow, e->d_prep, e->d_dyn, e->d_coop, e->d_wave, e->d_tab, e->d_ray, e->d_env_off, e->d_hf, e->d_motion_off, e->d_records, e->d_meta, e->d_weights, e->d_fail,
                    e->d_cdf, e->d_ema, e->d_done_list, e->d_done_key, e->d_chunk_count, e->d_motion_done, e->d_reset_count, e->d_reset_calls,
                    e->d_scratch_jr};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (e->h_health) (void)hipHostFree(e->h_health);
    for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->tev) if (ev) (void)hipEventDestroy(ev);
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
}

static int sync_params(ParcEnv *e) { // StepParams travels by value with every launch; a captured graph holds a copy
    e->graph_dirty = true;
    return PARC_OK;
}

extern "C" int parc_env_create(const ParcEnvConfig *cfg, ParcEnv **out) {
    if (!cfg || !out) return fail(PARC_ERR_INVALID, "null argument");
    if (cfg->abi_version != PARC_ABI_VERSION || cfg->struct_size != sizeof(ParcEnvConfig))
        return fail(PARC_ERR_INVALID, "ParcEnvConfig ABI mismatch (abi_version / struct_size)");
    const ParcCharModel &m = cfg->model;
    if (m.num_bodies < 2 || m.num_bodies > 15) return fail(PARC_ERR_INVALID, "num_bodies must be in [2,15]");
    if (m.dof_size < 1 || m.dof_size > PARC_MAX_DOFS) return fail(PARC_ERR_INVALID, "dof_size must be in [1,40]");
    if (cfg->num_tar_obs_steps < 1 || cfg->num_tar_obs_steps > PARC_MAX_TAR_STEPS) return fail(PARC_ERR_INVALID, "tar_obs_steps: 1..6 entries");
    if (cfg->num_key_bodies < 0 || cfg->num_key_bodies > PARC_MAX_KEY_BODIES) return fail(PARC_ERR_INVALID, "key_bodies: at most 8");
    if (cfg->num_envs < 1) return fail(PARC_ERR_INVALID, "num_envs must be >= 1");
    if (cfg->num_rays < 1 || cfg->num_rays > 4096 || !cfg->ray_points_host) return fail(PARC_ERR_INVALID, "ray fan: 1..4096 points");
    if (!cfg->env_offsets_host) return fail(PARC_ERR_INVALID, "env_offsets_host is required");
    for (int b = 1; b < m.num_bodies; ++b)
        if (m.parent[b] < 0 || m.parent[b] >= b) return fail(PARC_ERR_INVALID, "bodies must be in DFS order (parent < child)");
    if (cfg->enable_dynamics && !cfg->body_pos_from_fk)
        return fail(PARC_ERR_INVALID, "enable_dynamics needs body_pos_from_fk = 1 (the simulator is reduced-coordinate)");
    if (cfg->enable_dynamics && (cfg->dynamics.num_geoms < 1 || cfg->dynamics.num_geoms > PARC_MAX_GEOMS || !(cfg->dynamics.sim_dt > 0.f)))
        return fail(PARC_ERR_INVALID, "enable_dynamics: bad ParcDynamicsParams");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PARC_ERR_NO_DEVICE, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(PARC_ERR_INVALID, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));

    ParcEnv *e = new (std::nothrow) ParcEnv();
    if (!e) return fail(PARC_ERR_INVALID, "out of host memory");
    e->cfg = *cfg;
    e->dev_options_s = cfg->dev_options ? cfg->dev_options : "";
    e->cfg.dev_options = e->dev_options_s.c_str();
    e->cfg.ray_points_host = nullptr; e->cfg.env_offsets_host = nullptr;
    e->B = m.num_bodies; e->J = e->B - 1; e->D = m.dof_size; e->K = cfg->num_key_bodies; e->S = cfg->num_tar_obs_steps;
    e->R = cfg->num_rays; e->N = cfg->num_envs;
    const int B = e->B, J = e->J, D = e->D, K = e->K, S = e->S, R = e->R;

    DevTables &t = e->h_tab;
    memset(&t, 0, sizeof(t));
    for (int b = 0; b < 16; ++b) { t.h.parent[b] = -1; t.h.jtype[b] = PARC_JOINT_FIXED; t.h.key_slot[b] = -1; }
    for (int b = 0; b < B; ++b) {
        t.h.parent[b] = m.parent[b]; t.h.jtype[b] = m.joint_type[b]; t.h.dof_idx[b] = m.dof_idx[b];
        for (int c = 0; c < 3; ++c) { t.h.axis[b][c] = m.joint_axis[b][c]; t.h.lt[b][c] = m.local_translation[b][c]; }
        for (int c = 0; c < 4; ++c) t.h.lr[b][c] = m.local_rotation[b][c];
    }
    for (int p = 0; p < PARC_MAX_FK_PATHS; ++p)
        for (int d = 0; d < PARC_MAX_FK_DEPTH; ++d) {
            t.h.fk_paths[p][d] = m.fk_paths[p][d];
            if (m.fk_paths[p][d] >= B) { delete e; return fail(PARC_ERR_INVALID, "fk_paths entry out of range"); }
        }
    const float dt_f = (float)cfg->control_dt;
    for (int s = 0; s < S; ++s) t.tstep[s] = dt_f * (float)cfg->tar_obs_steps[s];
    for (int k = 0; k < K; ++k) {
        if (cfg->key_body_ids[k] < 0 || cfg->key_body_ids[k] >= B) { delete e; return fail(PARC_ERR_INVALID, "key body id out of range"); }
        t.h.key_ids[k] = cfg->key_body_ids[k];
        if (t.h.key_slot[cfg->key_body_ids[k]] < 0) t.h.key_slot[cfg->key_body_ids[k]] = k;
    }
    for (int j = 0; j < J; ++j) { t.joint_err_w[j] = cfg->joint_err_w[j]; t.pose_term_dist[j] = cfg->pose_termination_dist[j]; }
    for (int d = 0; d < D; ++d) t.dof_err_w[d] = cfg->dof_err_w[d];
    for (int b = 0; b < B; ++b) t.contact_w[b] = cfg->contact_weights[b];

    StepParams &sp = e->sp;
    memset(&sp, 0, sizeof(sp));
    sp.N = e->N; sp.B = B; sp.J = J; sp.D = D; sp.K = K; sp.S = S; sp.R = R;
    // observation layout (ig_parkour_env.py:842-965)
    sp.global_obs = cfg->global_obs; sp.off_char = cfg->global_root_height_obs ? 1 : 0;
    sp.off_dofvel = sp.off_char + 12 + 6 * J;
    sp.off_key = sp.off_dofvel + D;
    sp.off_tar = sp.off_key + 3 * K;
    sp.tar_w = 3 + 6 + 6 * J + 3 * K;
    sp.lr_identity = 1;
    for (int b = 0; b < B; ++b)
        if (!(cfg->model.local_rotation[b][0] == 0.f && cfg->model.local_rotation[b][1] == 0.f && cfg->model.local_rotation[b][2] == 0.f && cfg->model.local_rotation[b][3] == 1.f)) sp.lr_identity = 0;
    sp.obs_tar = cfg->enable_tar_obs != 0; sp.obs_contact = cfg->use_contact_info != 0;
    sp.off_tarc = sp.off_tar + (sp.obs_tar ? S * sp.tar_w : 0);          // ig_parkour_env.py:927-946: the blocks that exist, in this order
    sp.off_cc = sp.off_tarc + (sp.obs_tar && sp.obs_contact ? S * B : 0);
    sp.off_hf = sp.off_cc + (sp.obs_contact ? B : 0);
    sp.obs_dim = sp.off_hf + R;
    e->obs_dim = sp.obs_dim;
    for (int q = 0; q < S; ++q) sp.tstep[q] = dt_f * (float)cfg->tar_obs_steps[q];
    sp.dt_f = dt_f; sp.episode_length = cfg->episode_length; sp.min_obs_h = cfg->min_obs_h; sp.max_obs_h = cfg->max_obs_h;
    sp.pose_w = cfg->pose_w; sp.vel_w = cfg->vel_w; sp.root_pos_w = cfg->root_pos_w; sp.root_vel_w = cfg->root_vel_w; sp.key_pos_w = cfg->key_pos_w;
    sp.root_pos_term_sq = (float)((double)cfg->root_pos_termination_dist * (double)cfg->root_pos_termination_dist);
    sp.root_rot_term = cfg->root_rot_termination_angle;
    sp.early_term = cfg->enable_early_termination; sp.pose_term = cfg->pose_termination; sp.track_root = cfg->track_root;
    sp.track_root_h = cfg->track_root_h; sp.tracking = cfg->report_tracking_error; sp.body_pos_from_fk = cfg->body_pos_from_fk;
    sp.fall_mask = cfg->contact_body_mask & ((1u << B) - 1u); sp.term_h = cfg->termination_height;

    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t r = hipMalloc(dst, bytes);
        if (r != hipSuccess) return r;
        return src ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipMemset(*dst, 0, bytes);
    };
    hipError_t r = hipSuccess;
    const size_t N = (size_t)e->N;
    if ((r = up((void **)&e->d_tab, &t, sizeof(t))) != hipSuccess ||
        (r = up((void **)&e->d_prep, nullptr, sizeof(float4) * 16 * N)) != hipSuccess ||
        (r = up((void **)&e->d_ray, cfg->ray_points_host, sizeof(float) * 2 * R)) != hipSuccess ||
        (r = up((void **)&e->d_env_off, cfg->env_offsets_host, sizeof(float) * 3 * N)) != hipSuccess ||
        (r = up((void **)&e->d_ema, nullptr, ((N + 1023) / 1024) * 1024)) != hipSuccess ||
        (r = up((void **)&e->d_done_list, nullptr, sizeof(int) * N)) != hipSuccess ||
        (r = up((void **)&e->d_done_key, nullptr, sizeof(int) * N)) != hipSuccess ||
        (r = up((void **)&e->d_chunk_count, nullptr, sizeof(int) * 1024)) != hipSuccess ||
        (r = up((void **)&e->d_reset_count, nullptr, 2 * sizeof(int))) != hipSuccess ||
        (r = up((void **)&e->d_reset_calls, nullptr, sizeof(unsigned long long))) != hipSuccess ||
        (r = up((void **)&e->d_scratch_jr, nullptr, sizeof(float) * 4 * J * N)) != hipSuccess) {
        free_dev(e); delete e;
        return fail(PARC_ERR_HIP, std::string("device allocation failed: ") + hipGetErrorString(r));
    }
    {
        hipDeviceProp_t prop;
        (void)hipGetDeviceProperties(&prop, cfg->device);
        e->grid_waves = prop.multiProcessorCount * 16; // persistent waves; each strides over envs
        e->num_cus = prop.multiProcessorCount;
    }
    if (cfg->enable_dynamics) {
        memset(&e->h_dyn, 0, sizeof(e->h_dyn));
        parcdyn::fill_dyn_model(e->h_dyn, cfg->model, cfg->dynamics, cfg->action_low, cfg->action_high);
        if (e->h_dyn.truncated > 0 || cfg->dynamics.num_geoms > PARC_MAX_GEOMS) { // a silently shortened collision set would depend on the geom order
            const int t = e->h_dyn.truncated;
            free_dev(e); delete e;
            return fail(PARC_ERR_INVALID, "dynamics: " + std::to_string(t) + " collision point(s) / segment(s) / geom(s) do not fit the model tables (DYN_MAXC " +
                        std::to_string(DYN_MAXC) + ", DYN_MAXS " + std::to_string(DYN_MAXS) + ", " + std::to_string(PARC_MAX_GEOMS) + " geoms)");
        }
        { // control mode (ig_char_env.py:21-26, 95)
            const int cm = cfg->dynamics.control_mode;
            bool all_1d = true;
            for (int b = 1; b < cfg->model.num_bodies; ++b) all_1d = all_1d && cfg->model.joint_type[b] != parcdyn::DJ_SPHERICAL;
            if (cm < PARC_CTRL_PD || cm > PARC_CTRL_PD_1D || (cm == PARC_CTRL_PD_1D && !all_1d)) {
                free_dev(e); delete e;
                return fail(PARC_ERR_INVALID, cm == PARC_CTRL_PD_1D ? "control_mode pd_1d only supports characters whose joints all have one dof (the reference asserts it, ig_char_env.py:246-250)"
                                                                     : "unknown control_mode " + std::to_string(cm));
            }
        }
        // developer switches for ablation measurements (ParcEnvConfig::dev_options; the product never sets them)
        {
            const std::string mode = dev_opt(cfg, "segments"); // "none": collision points only; "capsules": drop the sole edges of boxes
            if (!mode.empty()) {
                parcdyn::DynModel &dm = e->h_dyn;
                int keep = 0;
                for (int k = 0; k < dm.nseg; ++k) {
                    const bool drop = mode == "none" || (mode == "capsules" && dm.seg_r[k] <= 0.011f);
                    if (drop) continue;
                    dm.seg_body[keep] = dm.seg_body[k]; dm.seg_r[keep] = dm.seg_r[k];
                    for (int a = 0; a < 3; ++a) { dm.seg_a[keep][a] = dm.seg_a[k][a]; dm.seg_b[keep][a] = dm.seg_b[k][a]; }
                    ++keep;
                }
                dm.nseg = keep;
            }
            const std::string dt_ = dev_opt(cfg, "dtang");
            if (!dt_.empty()) e->h_dyn.dtang = (float)atof(dt_.c_str());
            const std::string mp_ = dev_opt(cfg, "man_period"); // contact discovery every n substeps (1 = every substep, the round-3 behaviour)
            if (!mp_.empty() && atoi(mp_.c_str()) >= 1) {
                e->h_dyn.man_period = atoi(mp_.c_str());
                e->h_dyn.spec_tv = 1.5f * (float)(e->h_dyn.man_period - 1) * e->h_dyn.dt;
            }
        }
        if ((r = up((void **)&e->d_dyn, &e->h_dyn, sizeof(e->h_dyn))) != hipSuccess) {
            free_dev(e); delete e;
            return fail(PARC_ERR_HIP, std::string("device allocation failed: ") + hipGetErrorString(r));
        }
        // Kernel choice by tree shape: wave-per-limb (a trunk chain + <= 4 limb chains of <= 3 bodies, the humanoid),
        // else chain-parallel (<= 8 chains of <= 4 bodies), else thread-per-env.  PARC_DYN_KERNEL=coop|thread forces
        // one of the more general kernels (they are kept as fallbacks for other trees and as cross-checks).
        const std::string want = dev_opt(cfg, "kernel");
        const bool coop_ok = parcdyn::build_coop_tables(e->h_dyn, e->h_coop);
        const bool wave_ok = coop_ok && parcdyn::build_wave_tables(e->h_dyn, e->h_coop, e->h_wave);
        e->wave_rejected = !wave_ok; // the tree / the collision tables do not fit k_dynamics_wave (parc_env_describe says so: the fallbacks are several times slower)
        e->use_wave = wave_ok && want != "coop" && want != "thread";
        e->use_coop = coop_ok && !e->use_wave && want != "thread";
        if (e->use_wave) {
            r = up((void **)&e->d_wave, &e->h_wave, sizeof(e->h_wave));
            if (r == hipSuccess && dev_opt(cfg, "no_residual").empty()) { // (developer switch: the precision test measures the drift without it)
                r = hipMalloc((void **)&e->d_root_shadow, sizeof(float) * 6 * N);
                if (r == hipSuccess) r = hipMemset(e->d_root_shadow, 0xff, sizeof(float) * 6 * N); // NaN: matches no buffer value
            }
            if (r == hipSuccess) { // overflow area of the contact-plane lists (never read before it is written)
                const int epb = wave_envs_per_block(N, e->num_cus);
                r = hipMalloc((void **)&e->d_man_ovf, sizeof(float) * (size_t)((N + epb - 1) / epb) * WV_MAXLIMB * WV_MAN_OVF * 8 * 64);
            }
            if (r == hipSuccess)
                r = hipFuncSetAttribute(cfg->dynamics.control_mode == PARC_CTRL_PD ? (const void *)parcdyn::k_dynamics_wave : (const void *)parcdyn::k_dynamics_wave_ff,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, parcdyn::wv_lds_floats() * (int)sizeof(float));
        } else if (e->use_coop) {
            r = up((void **)&e->d_coop, &e->h_coop, sizeof(e->h_coop));
        }
        if (r != hipSuccess) {
            free_dev(e); delete e;
            return fail(PARC_ERR_HIP, std::string("dynamics kernel setup failed: ") + hipGetErrorString(r));
        }
    }
    e->force_ema_leader = !dev_opt(cfg, "ema_leader").empty();
    e->force_two_launch_curriculum = !dev_opt(cfg, "curriculum_two_launches").empty();
    if (hipHostMalloc((void **)&e->h_health, 2 * sizeof(unsigned int), hipHostMallocMapped) == hipSuccess) {
        e->h_health[0] = 0u; e->h_health[1] = 0u;
        if (hipHostGetDevicePointer((void **)&e->d_health, e->h_health, 0) != hipSuccess) e->d_health = nullptr;
    }
    for (auto &ev : e->ev) (void)hipEventCreate(&ev);
    sp.tables = e->d_tab; sp.ray_points = e->d_ray; sp.env_offsets = e->d_env_off;
    sp.ema_code = e->d_ema; sp.prep = e->d_prep;
#ifdef PARC_STAMPS
    if (hipMalloc((void **)&sp.stamp_out, sizeof(unsigned) * 8 * N) != hipSuccess) sp.stamp_out = nullptr;
#endif
    e->nchunks = (e->N + 1023) / 1024;
    if (e->nchunks > 1024) { free_dev(e); delete e; return fail(PARC_ERR_INVALID, "num_envs must be <= 1048576 per handle"); }

    *out = e;
    return PARC_OK;
}

extern "C" void parc_env_destroy(ParcEnv *e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    free_dev(e);
    delete e;
}

extern "C" int parc_env_obs_dim(const ParcEnv *e) { return e ? e->obs_dim : PARC_ERR_INVALID; }

extern "C" int parc_env_load_motions(ParcEnv *e, const ParcMotionClips *c) {
    if (!e || !c || c->num_motions < 1) return fail(PARC_ERR_INVALID, "bad motion clips");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int M = c->num_motions, B = e->B, J = e->J;
    int64_t F = 0;
    e->h_meta.assign(M, MotionMeta());
    e->h_weights.assign(M, 0.f);
    float wsum = 0.f;
    for (int m = 0; m < M; ++m) {
        if (c->num_frames_host[m] < 2) return fail(PARC_ERR_INVALID, "every clip needs at least 2 frames");
        if (c->fps_host[m] <= 0) return fail(PARC_ERR_INVALID, "fps must be positive");
        if (c->weights_host[m] < 0) return fail(PARC_ERR_INVALID, "motion weights must be >= 0");
        e->h_weights[m] = (float)c->weights_host[m];
        wsum = wsum + e->h_weights[m];
    }
    std::vector<int> frame_motion;
    for (int m = 0; m < M; ++m) {
        MotionMeta &mm = e->h_meta[m];
        const int n = c->num_frames_host[m];
        mm.start = (int)F; mm.nframes = n; mm.loop = c->loop_modes_host[m]; mm.fps = (float)c->fps_host[m];
        mm.length = (float)(1.0 / (double)c->fps_host[m] * (double)(n - 1)); // motion_lib.py:305
        const float *rp = c->root_pos_host + 3 * F;
        mm.dx = rp[3 * (n - 1)] - rp[0]; mm.dy = rp[3 * (n - 1) + 1] - rp[1]; mm.dz = 0.f; // :307-308
        e->h_weights[m] = e->h_weights[m] / wsum; // :372
        for (int f = 0; f < n; ++f) frame_motion.push_back(m);
        F += n;
    }
    if (F > (int64_t)1 << 30) return fail(PARC_ERR_INVALID, "too many frames");
    void *olds[] = {e->d_records, e->d_meta, e->d_weights, e->d_fail, e->d_cdf, e->d_motion_done};
    for (void *p : olds) if (p) (void)hipFree(p);
    e->d_records = nullptr; e->d_meta = nullptr; e->d_weights = nullptr; e->d_fail = nullptr; e->d_cdf = nullptr; e->d_motion_done = nullptr;
    float *d_rp = nullptr, *d_rr = nullptr, *d_jr = nullptr, *d_ct = nullptr;
    int *d_fm = nullptr;
    HIPCHK(hipMalloc((void **)&e->d_records, sizeof(float4) * REC_F4 * F));
    HIPCHK(hipMalloc((void **)&e->d_meta, sizeof(MotionMeta) * M));
    HIPCHK(hipMalloc((void **)&e->d_weights, sizeof(float) * M));
    HIPCHK(hipMalloc((void **)&e->d_fail, sizeof(float) * M));
    HIPCHK(hipMalloc((void **)&e->d_cdf, sizeof(float) * M));
    HIPCHK(hipMalloc((void **)&e->d_motion_done, sizeof(int) * M));
    HIPCHK(hipMemset(e->d_motion_done, 0x7f, sizeof(int) * M)); // 0x7f7f7f7f: larger than any list index (k_ema_first takes the minimum)
    HIPCHK(hipMalloc((void **)&d_rp, sizeof(float) * 3 * F));
    HIPCHK(hipMalloc((void **)&d_rr, sizeof(float) * 4 * F));
    HIPCHK(hipMalloc((void **)&d_jr, sizeof(float) * 4 * J * F));
    HIPCHK(hipMalloc((void **)&d_fm, sizeof(int) * F));
    HIPCHK(hipMemcpy(e->d_meta, e->h_meta.data(), sizeof(MotionMeta) * M, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_weights, e->h_weights.data(), sizeof(float) * M, hipMemcpyHostToDevice));
    std::vector<float> ones(M, 1.0f); // dm_env.py:87
    HIPCHK(hipMemcpy(e->d_fail, ones.data(), sizeof(float) * M, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_rp, c->root_pos_host, sizeof(float) * 3 * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_rr, c->root_rot_host, sizeof(float) * 4 * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_jr, c->joint_rot_host, sizeof(float) * 4 * J * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_fm, frame_motion.data(), sizeof(int) * F, hipMemcpyHostToDevice));
    if (c->contacts_host) {
        HIPCHK(hipMalloc((void **)&d_ct, sizeof(float) * B * F));
        HIPCHK(hipMemcpy(d_ct, c->contacts_host, sizeof(float) * B * F, hipMemcpyHostToDevice));
    }
    PrepParams pp;
    pp.F = (int)F; pp.B = B; pp.J = J; pp.D = e->D; pp.root_pos = d_rp; pp.root_rot = d_rr; pp.joint_rot = d_jr; pp.contacts = d_ct;
    pp.frame_motion = d_fm; pp.meta = e->d_meta; pp.tables = e->d_tab; pp.records = e->d_records;
    hipLaunchKernelGGL(k_motion_prep, dim3((unsigned)((F + 127) / 128)), dim3(128), 0, 0, pp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(d_rp); (void)hipFree(d_rr); (void)hipFree(d_jr); (void)hipFree(d_fm);
    if (d_ct) (void)hipFree(d_ct);
    e->M = M; e->F = F; e->have_motions = true;
    e->sp.M = M; e->sp.records = e->d_records; e->sp.meta = e->d_meta;
    return sync_params(e);
}

extern "C" int parc_env_load_terrain(ParcEnv *e, const float *hf, int32_t X, int32_t Y, float min_x, float min_y, float dx, float dy,
                                     const float *motion_offsets, int32_t M, int32_t T) {
    if (!e || !hf || !motion_offsets || X < 1 || Y < 1 || T < 1 || !(dx > 0.f) || !(dy > 0.f)) return fail(PARC_ERR_INVALID, "bad terrain");
    if (!e->have_motions || M != e->M) return fail(PARC_ERR_STATE, "load_motions first; motion_offsets must have one row per motion");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (e->d_hf) (void)hipFree(e->d_hf);
    if (e->d_motion_off) (void)hipFree(e->d_motion_off);
    e->d_hf = nullptr; e->d_motion_off = nullptr;
    HIPCHK(hipMalloc((void **)&e->d_hf, sizeof(float) * (size_t)X * Y));
    HIPCHK(hipMalloc((void **)&e->d_motion_off, sizeof(float) * 2 * (size_t)M * T));
    HIPCHK(hipMemcpy(e->d_hf, hf, sizeof(float) * (size_t)X * Y, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_motion_off, motion_offsets, sizeof(float) * 2 * (size_t)M * T, hipMemcpyHostToDevice));
    StepParams &sp = e->sp;
    sp.hf = e->d_hf; sp.X = X; sp.Y = Y; sp.min_x = min_x; sp.min_y = min_y; sp.dx = dx; sp.dy = dy; sp.T = T;
    sp.rdx = (float)(1.0L / (long double)dx); sp.rdy = (float)(1.0L / (long double)dy);
    sp.motion_offsets = e->d_motion_off;
    e->T = T;
    // terrain tile radius: farthest ray sample in cells, +1 for the two independent roundings
    std::vector<float> ray(2 * (size_t)e->R);
    HIPCHK(hipMemcpy(ray.data(), e->d_ray, sizeof(float) * 2 * e->R, hipMemcpyDeviceToHost));
    float rmax = 0.f;
    for (int r = 0; r < e->R; ++r) rmax = fmaxf(rmax, sqrtf(ray[2 * r] * ray[2 * r] + ray[2 * r + 1] * ray[2 * r + 1]));
    int tr = (int)ceilf(rmax / fminf(dx, dy) + 0.01f); // max |round(a+d)-round(a)| = ceil(|d|)
    if ((2 * tr + 1) * (2 * tr + 1) > TILE_MAX_CELLS) tr = -1; // fan too wide for the LDS tile: direct gathers
    sp.tile_r = tr;
    if (tr >= 0) { // magic multiplier for idx / TW, verified for every cell of the tile
        const unsigned TW = 2 * tr + 1;
        const unsigned mul = (65536u + TW - 1) / TW;
        for (unsigned idx = 0; idx < TW * TW; ++idx)
            if (((idx * mul) >> 16) != idx / TW) return fail(PARC_ERR_INVALID, "internal: tile index multiplier");
        sp.tile_mul = mul;
    }
    const int stage_pad = (sp.off_tarc + 3) & ~3;
    e->lds_bytes = sizeof(float) * (size_t)stage_pad + (e->cfg.report_tracking_error ? 2 * 16 * sizeof(float4) : 0); // per wave of k_env_post
    e->sp.lds_wave_floats = (int)(e->lds_bytes / sizeof(float));
    e->have_terrain = true;
    e->graph_dirty = true;
    return sync_params(e);
}

extern "C" int parc_env_bind_buffers(ParcEnv *e, const ParcEnvBuffers *b) {
    if (!e || !b) return fail(PARC_ERR_INVALID, "null argument");
    const void *req[] = {b->char_root_pos, b->char_root_rot, b->char_root_vel, b->char_root_ang_vel, b->char_dof_pos, b->char_dof_vel,
                         b->contact_forces, b->motion_ids, b->terrain_ids, b->time_offsets, b->timestep, b->obs, b->reward, b->done};
    for (const void *p : req) if (!p) return fail(PARC_ERR_INVALID, "a required buffer is NULL");
    if (!e->cfg.body_pos_from_fk && !b->char_body_pos) return fail(PARC_ERR_INVALID, "char_body_pos is required when body_pos_from_fk == 0");
    if (((uintptr_t)b->obs & 15) != 0) return fail(PARC_ERR_INVALID, "obs must be 16-byte aligned");
    if ((b->ref_root_rot && ((uintptr_t)b->ref_root_rot & 15)) || (b->ref_joint_rot && ((uintptr_t)b->ref_joint_rot & 15)) ||
        ((uintptr_t)b->char_root_rot & 15))
        return fail(PARC_ERR_INVALID, "quaternion buffers must be 16-byte aligned");
    e->sp.buf = *b;
    e->bound = true;
    e->graph_dirty = true;
    return sync_params(e);
}

static int check_ready(ParcEnv *e) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    if (!e->bound || !e->have_motions || !e->have_terrain) return fail(PARC_ERR_STATE, "bind_buffers, load_motions and load_terrain must precede this call");
    return PARC_OK;
}

static int launch_dynamics(ParcEnv *e, const float *action_dev, hipStream_t st) {
    if (!e->cfg.enable_dynamics) return PARC_OK;
    if (!action_dev) return fail(PARC_ERR_INVALID, "action is required when enable_dynamics is set");
    parcdyn::DynTerrain T;
    T.hf = e->d_hf; T.X = e->sp.X; T.Y = e->sp.Y; T.min_x = e->sp.min_x; T.min_y = e->sp.min_y; T.dx = e->sp.dx; T.dy = e->sp.dy;
    if (e->use_wave) {
        const int epb = wave_envs_per_block(e->N, e->num_cus);
        if (e->cfg.dynamics.control_mode == PARC_CTRL_PD)
            hipLaunchKernelGGL(parcdyn::k_dynamics_wave, dim3((e->N + epb - 1) / epb), dim3(256), parcdyn::wv_lds_floats() * sizeof(float), st,
                               (const parcdyn::DynModel *)e->d_dyn, (const parcdyn::WaveTables *)e->d_wave, T, e->sp.buf, action_dev,
                               (const float *)e->d_env_off, e->d_root_shadow, e->d_prep, e->d_man_ovf, e->N, epb);
        else // control modes vel / torque / pd_exp / pd_1d: the instantiation with the feed-forward joint torques
            hipLaunchKernelGGL(parcdyn::k_dynamics_wave_ff, dim3((e->N + epb - 1) / epb), dim3(256), parcdyn::wv_lds_floats() * sizeof(float), st,
                               (const parcdyn::DynModel *)e->d_dyn, (const parcdyn::WaveTables *)e->d_wave, T, e->sp.buf, action_dev,
                               (const float *)e->d_env_off, e->d_root_shadow, e->d_prep, e->d_man_ovf, e->N, epb);
    }
    else if (e->use_coop)
        hipLaunchKernelGGL(parcdyn::k_dynamics_coop, dim3((e->N + CO_ENVS - 1) / CO_ENVS), dim3(64), 0, st, (const parcdyn::DynModel *)e->d_dyn,
                           (const parcdyn::CoopTables *)e->d_coop, T, e->sp.buf, action_dev, (const float *)e->d_env_off, e->N);
    else
        hipLaunchKernelGGL(k_dynamics, dim3((e->N + 63) / 64), dim3(64), 0, st, (const parcdyn::DynModel *)e->d_dyn, T, e->sp.buf, action_dev,
                           (const float *)e->d_env_off, e->N);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

static int launch_curriculum(ParcEnv *e, hipStream_t st) {
    e->done_list_fresh = true;
    if (e->M <= 64 && !e->force_ema_leader && e->N <= CURRICULUM_ONE_LAUNCH_MAX && (e->N & 7) == 0 && !e->force_two_launch_curriculum) {
        hipLaunchKernelGGL(k_curriculum_small, dim3(e->nchunks + e->M), dim3(1024), 0, st, e->d_ema, e->sp.buf.motion_ids, e->N, e->d_done_list,
                           e->d_done_key, e->d_reset_count, e->sp.never_done, e->nchunks, e->d_fail, e->M, e->cfg.fail_rate_ema_weight, e->d_health);
        HIPCHK(hipGetLastError());
        return PARC_OK;
    }
    hipLaunchKernelGGL(k_done_scatter, dim3(e->nchunks), dim3(1024), 0, st, e->d_ema, e->sp.buf.motion_ids, e->N, e->d_done_list,
                       e->d_done_key, e->d_reset_count, e->sp.never_done, e->d_health);
    if (e->M <= 64 && !e->force_ema_leader) {
        hipLaunchKernelGGL(k_fail_rate_ema, dim3(e->M), dim3(1024), 0, st, e->d_done_key, e->d_reset_count, e->d_fail, e->M,
                           e->cfg.fail_rate_ema_weight);
    } else { // d_motion_done doubles as the per-motion "first list entry" slot (INT_MAX between steps)
        hipLaunchKernelGGL(k_ema_first, dim3((e->N + 255) / 256), dim3(256), 0, st, e->d_done_key, e->d_reset_count, e->d_motion_done);
        hipLaunchKernelGGL(k_ema_leader, dim3((e->N + 3) / 4), dim3(256), 0, st, e->d_done_key, e->d_reset_count, e->d_motion_done, e->d_fail,
                           e->cfg.fail_rate_ema_weight);
    }
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

// an observation layout other than the default one: the OBSVAR instantiations of k_env_post decide it at run time
static bool obs_variant(const ParcEnvConfig &c) { return c.global_obs || c.global_root_height_obs || !c.use_contact_info || !c.enable_tar_obs; }

static bool wants_mirror(const ParcEnvBuffers &b) {
    return b.ref_root_pos || b.ref_root_rot || b.ref_root_vel || b.ref_root_ang_vel || b.ref_joint_rot || b.ref_dof_pos || b.ref_dof_vel ||
           b.ref_body_pos || b.ref_contacts || b.ray_hfs || b.tracking_error;
}

static int launch_post(ParcEnv *e, int mode, const int64_t *ids, int count, hipStream_t st, const int *ids32 = nullptr,
                       const int *count_dev = nullptr, bool prep_done = false, unsigned long long *bump = nullptr) {
    if (count <= 0) return PARC_OK;
    const int grid = (count + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    const size_t lds = e->lds_bytes * ENVS_PER_BLOCK;
    if (mode == MODE_STEP && e->cfg.enable_dynamics && e->use_wave) prep_done = true; // k_dynamics_wave wrote the prep records with the state
    if (!prep_done) hipLaunchKernelGGL(k_env_prep, dim3((count + 3) / 4), dim3(64), 0, st, e->sp, ids, ids32, count_dev, count);
    const bool mirror = wants_mirror(e->sp.buf);
    if (obs_variant(e->cfg)) { // a non-default observation layout -- instantiated for the general (MIRROR) form only
        if (mode == MODE_STEP && !e->sp.track_root) hipLaunchKernelGGL((k_env_post<MODE_STEP, true, true, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
        else if (mode == MODE_STEP) hipLaunchKernelGGL((k_env_post<MODE_STEP, true, false, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
        else hipLaunchKernelGGL((k_env_post<MODE_OBS, true, false, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
    } else if (mode == MODE_STEP && !e->sp.track_root) { // track_root: false -- the reward in the characters' heading frames
        if (mirror) hipLaunchKernelGGL((k_env_post<MODE_STEP, true, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
        else hipLaunchKernelGGL((k_env_post<MODE_STEP, false, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
    } else if (mode == MODE_STEP) {
        if (mirror) hipLaunchKernelGGL((k_env_post<MODE_STEP, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
        else hipLaunchKernelGGL((k_env_post<MODE_STEP, false>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
    } else {
        if (mirror) hipLaunchKernelGGL((k_env_post<MODE_OBS, true>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
        else hipLaunchKernelGGL((k_env_post<MODE_OBS, false>), dim3(grid), dim3(64 * ENVS_PER_BLOCK), lds, st, e->sp, ids, ids32, count_dev, count, bump);
    }

    HIPCHK(hipGetLastError());
    return PARC_OK;
}

extern "C" int parc_env_step(ParcEnv *e, const float *action_dev, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t *tv = nullptr;
    if (e->timing) { // no host synchronisation: events come from a growing pool and are read back by parc_env_get_kernel_timing
        if (e->tev_used + 4 > e->tev.size()) {
            const size_t old = e->tev.size();
            e->tev.resize(old + 1024, nullptr);
            for (size_t i = old; i < e->tev.size(); ++i) HIPCHK(hipEventCreate(&e->tev[i]));
        }
        tv = &e->tev[e->tev_used];
        e->tev_used += 4;
        HIPCHK(hipEventRecord(tv[0], st));
    }
    rc = launch_dynamics(e, action_dev, st);
    if (rc) return rc;
    if (tv) HIPCHK(hipEventRecord(tv[1], st));
    rc = launch_post(e, MODE_STEP, nullptr, e->N, st);
    if (rc) return rc;
    if (tv) HIPCHK(hipEventRecord(tv[2], st));
    rc = launch_curriculum(e, st);
    if (tv) HIPCHK(hipEventRecord(tv[3], st));
    return rc;
}

extern "C" int parc_env_get_buffers(ParcEnv *e, ParcEnvBuffers *out) {
    if (!e || !out) return fail(PARC_ERR_INVALID, "null argument");
    if (!e->bound) return fail(PARC_ERR_STATE, "bind_buffers first");
    *out = e->sp.buf;
    return PARC_OK;
}

extern "C" int parc_env_set_episode_length(ParcEnv *e, float seconds) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    if (!(seconds > 0.f)) return fail(PARC_ERR_INVALID, "episode length must be positive");
    e->sp.episode_length = seconds;
    return sync_params(e);
}

__global__ void k_td_lambda(const float *__restrict__ r, const float *__restrict__ nv, const int *__restrict__ done, float g, float lam0, int T, int N,
                            float *__restrict__ ret) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    size_t o = (size_t)(T - 1) * N + e;
    float acc = r[o] + g * nv[o];
    ret[o] = acc;
    for (int i = T - 2; i >= 0; --i) {
        o -= N;
        const float reset = done[o] != PARC_DONE_NULL ? 1.0f : 0.0f;
        const float lam = lam0 * (1.0f - reset);
        acc = r[o] + g * ((1.0f - lam) * nv[o] + lam * acc);
        ret[o] = acc;
    }
}

extern "C" int parc_td_lambda_return(const float *reward, const float *next_vals, const int32_t *done, float discount, float td_lambda,
                                     int32_t T, int32_t N, float *ret_out, void *stream) {
    if (!reward || !next_vals || !done || !ret_out || T < 1 || N < 0) return fail(PARC_ERR_INVALID, "bad argument");
    if (N == 0) return PARC_OK;
    hipLaunchKernelGGL(k_td_lambda, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, reward, next_vals, done, discount, td_lambda, T, N, ret_out);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

__global__ __launch_bounds__(256) void k_normalize_record(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ sd,
                                                          float clip, float *__restrict__ norm_out, float *__restrict__ copy_out, long long total, int dim) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % dim);
    const float v = x[i];
    float y = (v - mean[c]) / sd[c];
    y = fminf(fmaxf(y, -clip), clip);
    norm_out[i] = y;
    if (copy_out) copy_out[i] = v;
}

extern "C" int parc_normalize_record(const float *x, const float *mean, const float *sd, float clip, float *norm_out, float *copy_out,
                                     int64_t n, int32_t dim, void *stream) {
    if (!x || !mean || !sd || !norm_out || n < 0 || dim < 1) return fail(PARC_ERR_INVALID, "bad argument");
    const long long total = (long long)n * dim;
    if (total == 0) return PARC_OK;
    hipLaunchKernelGGL(k_normalize_record, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mean, sd, clip, norm_out,
                       copy_out, total, dim);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

// Recorder: one 128-thread block per env appends that env's row (see include/parc_env.h).
// ---------------------------------------------------------------------------------------------------------------------------------
// Minibatch gather (experience_buffer.py:81-89): block = 256 sampled rows.  Narrow buffers (a scalar or a few values per sample: reward,
// done, advantage ...): thread t copies row t.  Wide buffers (observations 5 248 B, actions 112 B): the block's four waves walk its 256 rows,
// a wavefront per row, 16 bytes per lane where source and destination rows are 16-byte aligned (else 4 / 1 bytes).  HBM-bound: 2 x the
// minibatch's bytes; bit-identical to index_select by construction (a byte copy).
struct GatherArgs {
    const char *src[PARC_MAX_GATHER_BUFFERS];
    char *dst[PARC_MAX_GATHER_BUFFERS];
    long long row_bytes[PARC_MAX_GATHER_BUFFERS];
    int nb;
};

__global__ __launch_bounds__(256) void k_gather_rows(const GatherArgs A, const long long *__restrict__ idx, long long n, long long count) {
    __shared__ long long s_row[256];
    const long long r0 = (long long)blockIdx.x * 256;
    const int t = threadIdx.x;
    long long mine = -1;
    if (r0 + t < n) {
        long long i = idx[r0 + t] % count;   // torch.remainder(idx, sample_count): idx >= 0
        if (i < 0) i += count;
        mine = i;
    }
    s_row[t] = mine;
    __syncthreads();
    const int lane = t & 63, w = t >> 6;
    for (int b = 0; b < A.nb; ++b) {
        const long long rb = A.row_bytes[b];
        if (rb < 64) { // narrow: one thread per row
            if (mine >= 0) {
                const char *s = A.src[b] + mine * rb;
                char *d = A.dst[b] + (r0 + t) * rb;
                if ((rb & 3) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 3) == 0) for (long long k = 0; k < rb; k += 4) *(int *)(d + k) = *(const int *)(s + k);
                else for (long long k = 0; k < rb; ++k) d[k] = s[k];
            }
        } else {       // wide: one wavefront per row, rows w, w + 4, ... of the block
            for (int q = w; q < 256; q += 4) {
                const long long i = s_row[q];
                if (i < 0) break; // rows past n (only at the end of the last block)
                const char *s = A.src[b] + i * rb;
                char *d = A.dst[b] + (r0 + q) * rb;
                if ((rb & 15) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
                    for (long long k = 16ll * lane; k < rb; k += 1024) *(float4 *)(d + k) = *(const float4 *)(s + k);
                } else if ((rb & 3) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 3) == 0) {
                    for (long long k = 4ll * lane; k < rb; k += 256) *(int *)(d + k) = *(const int *)(s + k);
                } else {
                    for (long long k = lane; k < rb; k += 64) d[k] = s[k];
                }
            }
        }
    }
}

extern "C" int parc_gather_rows(int32_t num_buffers, const void *const *src_dev, void *const *dst_dev, const int64_t *row_bytes, const int64_t *idx_dev,
                                int64_t n, int64_t count, void *stream) {
    if (num_buffers < 1 || num_buffers > PARC_MAX_GATHER_BUFFERS) return fail(PARC_ERR_INVALID, "parc_gather_rows: 1..16 buffers");
    if (!src_dev || !dst_dev || !row_bytes || !idx_dev || n < 0 || count < 1) return fail(PARC_ERR_INVALID, "parc_gather_rows: bad argument");
    if (n == 0) return PARC_OK;
    GatherArgs A;
    memset(&A, 0, sizeof(A));
    A.nb = num_buffers;
    for (int b = 0; b < num_buffers; ++b) {
        if (!src_dev[b] || !dst_dev[b] || row_bytes[b] < 1) return fail(PARC_ERR_INVALID, "parc_gather_rows: null buffer or empty row");
        A.src[b] = (const char *)src_dev[b]; A.dst[b] = (char *)dst_dev[b]; A.row_bytes[b] = row_bytes[b];
    }
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, (const long long *)idx_dev, (long long)n, (long long)count);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

__global__ __launch_bounds__(128) void k_record(const DevTables *__restrict__ T, ParcEnvBuffers buf, int N, int B, int D, int obs_dim, float *frames,
                                                float *obs_out, int cap, int *count, unsigned char *writing, int *n_writing, int use_ref) {
    const int e = blockIdx.x, tid = threadIdx.x;
    if (e >= N || !writing[e]) return;
    const int t = count[e];
    const int W = 3 + 4 + 4 * (B - 1) + B;
    if (t < cap) {
        float *row = frames + ((size_t)t * N + e) * W;
        const float *rp = use_ref ? buf.ref_root_pos : buf.char_root_pos, *rr = use_ref ? buf.ref_root_rot : buf.char_root_rot;
        if (tid < 3) row[tid] = rp[3 * (size_t)e + tid];
        if (tid < 4) row[3 + tid] = rr[4 * (size_t)e + tid];
        if (tid >= 1 && tid < B) {
            Q4 q;
            if (use_ref) q = *(const float4 *)(buf.ref_joint_rot + 4 * ((size_t)e * (B - 1) + tid - 1));
            else q = joint_dof_to_rot(T->h.jtype[tid], T->h.axis[tid], buf.char_dof_pos + (size_t)e * D + T->h.dof_idx[tid]);
            float *o = row + 7 + 4 * (tid - 1);
            o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
        }
        if (tid < B) {
            float c;
            if (use_ref) c = buf.ref_contacts[(size_t)e * B + tid];
            else {
                const float *f = buf.contact_forces + 3 * ((size_t)e * B + tid);
                c = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 1e-5f ? 1.f : 0.f;
            }
            row[7 + 4 * (B - 1) + tid] = c;
        }
        if (obs_out) {
            const float *src = buf.obs + (size_t)e * obs_dim;
            float *dst = obs_out + ((size_t)t * N + e) * obs_dim;
            for (int i = tid; i < obs_dim; i += 128) dst[i] = src[i];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const bool full = t >= cap;
        if (!full) count[e] = t + 1;
        if (buf.done[e] == PARC_DONE_FAIL || full) { // update_done ran before this: the row just written is the last one
            writing[e] = 0;
            atomicSub(n_writing, 1);
        }
    }
}

extern "C" int parc_env_record_bind(ParcEnv *e, float *frames_dev, float *obs_dev, int32_t cap, int32_t *count_dev, uint8_t *writing_dev,
                                    int32_t *n_writing_dev, int32_t record_ref) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    if (!frames_dev) { e->rec_frames = nullptr; e->rec_cap = 0; return PARC_OK; } // unbind
    if (cap < 1 || !count_dev || !writing_dev || !n_writing_dev) return fail(PARC_ERR_INVALID, "recorder buffers are incomplete");
    if (!e->bound) return fail(PARC_ERR_STATE, "bind_buffers must precede record_bind");
    if (record_ref && (!e->sp.buf.ref_root_pos || !e->sp.buf.ref_root_rot || !e->sp.buf.ref_joint_rot || !e->sp.buf.ref_contacts))
        return fail(PARC_ERR_STATE, "record_ref needs the ref_* mirrors to be bound");
    e->rec_frames = frames_dev; e->rec_obs = obs_dev; e->rec_cap = cap; e->rec_count = count_dev; e->rec_writing = writing_dev;
    e->rec_nwriting = n_writing_dev; e->rec_ref = record_ref ? 1 : 0;
    return PARC_OK;
}

extern "C" int parc_env_record_frame(ParcEnv *e, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->rec_frames) return fail(PARC_ERR_STATE, "no recorder buffers bound");
    hipLaunchKernelGGL(k_record, dim3(e->N), dim3(128), 0, (hipStream_t)stream, (const DevTables *)e->d_tab, e->sp.buf, e->N, e->B, e->D,
                       e->obs_dim, e->rec_frames, e->rec_obs, e->rec_cap, e->rec_count, e->rec_writing, e->rec_nwriting, e->rec_ref);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

extern "C" int parc_env_set_kernel_timing(ParcEnv *e, int32_t enable) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    e->timing = enable != 0;
    e->tev_used = 0;
    return PARC_OK;
}

// Per-step samples of the same events (call BEFORE parc_env_get_kernel_timing, which clears them): up to `cap` steps, dynamics kernel /
// observation kernel / the curriculum launches that close the step.  Returns the number of steps recorded in *steps.
extern "C" int parc_env_get_kernel_timing_samples(ParcEnv *e, float *dynamics_ms, float *obs_ms, float *curriculum_ms, int32_t cap, int32_t *steps) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    const size_t n = e->tev_used / 4;
    for (size_t i = 0; i < n && (int32_t)i < cap; ++i) {
        float a = 0.f, b = 0.f, c = 0.f;
        HIPCHK(hipEventSynchronize(e->tev[4 * i + 3]));
        HIPCHK(hipEventElapsedTime(&a, e->tev[4 * i], e->tev[4 * i + 1]));
        HIPCHK(hipEventElapsedTime(&b, e->tev[4 * i + 1], e->tev[4 * i + 2]));
        HIPCHK(hipEventElapsedTime(&c, e->tev[4 * i + 2], e->tev[4 * i + 3]));
        if (dynamics_ms) dynamics_ms[i] = a;
        if (obs_ms) obs_ms[i] = b;
        if (curriculum_ms) curriculum_ms[i] = c;
    }
    if (steps) *steps = (int32_t)n;
    return PARC_OK;
}

extern "C" int parc_env_get_kernel_timing(ParcEnv *e, double *dynamics_ms_avg, double *obs_ms_avg, int32_t *steps) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    const size_t n = e->tev_used / 4;
    double dyn = 0.0, post = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventSynchronize(e->tev[4 * i + 3]));
        HIPCHK(hipEventElapsedTime(&a, e->tev[4 * i], e->tev[4 * i + 1]));
        HIPCHK(hipEventElapsedTime(&b, e->tev[4 * i + 1], e->tev[4 * i + 2]));
        dyn += a; post += b;
    }
    if (dynamics_ms_avg) *dynamics_ms_avg = n ? dyn / (double)n : 0.0;
    if (obs_ms_avg) *obs_ms_avg = n ? post / (double)n : 0.0;
    if (steps) *steps = (int32_t)n;
    e->tev_used = 0;
    return PARC_OK;
}

extern "C" int parc_env_compute_obs(ParcEnv *e, const int64_t *ids, int32_t k, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (k == 0) return PARC_OK;
    if (k > 0 && !ids) return fail(PARC_ERR_INVALID, "env_ids is NULL");
    return launch_post(e, MODE_OBS, k < 0 ? nullptr : ids, k < 0 ? e->N : k, (hipStream_t)stream);
}

static ResetParams make_reset_params(ParcEnv *e) {
    ResetParams rp;
    rp.B = e->B; rp.J = e->J; rp.D = e->D; rp.T = e->T; rp.N = e->N;
    rp.records = e->d_records; rp.meta = e->d_meta; rp.motion_offsets = e->d_motion_off; rp.env_offsets = e->d_env_off;
    rp.tables = e->d_tab; rp.scratch_jr = e->d_scratch_jr; rp.prep = e->d_prep; rp.buf = e->sp.buf;
    return rp;
}

static SampleParams make_sample_params(ParcEnv *e, bool enabled) {
    SampleParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.enabled = enabled ? 1 : 0;
    sp.M = e->M; sp.T = e->T; sp.rand_reset = e->cfg.rand_reset; sp.demo_mode = e->cfg.demo_mode;
    sp.min_w = e->cfg.min_motion_weight; sp.noise_scale = e->cfg.rand_root_pos_offset_scale;
    sp.cdf_global = e->d_cdf; sp.fail_rates = e->d_fail; sp.motion_weights = e->d_weights; sp.start_frac = e->d_start_frac;
    sp.seed = (unsigned long long)e->cfg.seed; sp.call_dev = e->d_reset_calls;
    return sp;
}

extern "C" int parc_env_reset_with(ParcEnv *e, const int64_t *ids, int32_t k, const int32_t *mids, const int32_t *tids, const float *t0,
                                   const float *noise, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (k == 0) return PARC_OK;
    if (k > e->N) return fail(PARC_ERR_INVALID, "k > num_envs");
    if (!mids || !tids || !t0 || !noise || (k > 0 && !ids)) return fail(PARC_ERR_INVALID, "null sample array");
    const int n = k < 0 ? e->N : k;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_reset_with, dim3((n + 3) / 4), dim3(64), 0, st, make_reset_params(e), k < 0 ? nullptr : ids, (const int *)nullptr,
                       (const int *)nullptr, n, mids, tids, t0, noise, make_sample_params(e, false));
    HIPCHK(hipGetLastError());
    return launch_post(e, MODE_OBS, k < 0 ? nullptr : ids, n, st, nullptr, nullptr, /*prep_done=*/true); // the caller drew the samples: the sampling counter stays
}

// A sampling reset (dm_env.py:470-504 + :567-592): [k_build_cdf for libraries of more than CDF_LOCAL_MAX motions,] k_reset_with drawing
// motion / terrain / start time / xy noise per env, the observation pass (which bumps the Philox call index).
static int sampling_reset(ParcEnv *e, const int64_t *ids, const int *ids32, const int *count_dev, int n, hipStream_t st) {
    if (e->M > CDF_LOCAL_MAX)
        hipLaunchKernelGGL(k_build_cdf, dim3(1), dim3(1024), 0, st, e->d_fail, e->d_weights, e->cfg.min_motion_weight, e->M, e->d_cdf);
    hipLaunchKernelGGL(k_reset_with, dim3((n + 3) / 4), dim3(64), 0, st, make_reset_params(e), ids, ids32, count_dev, n, (const int *)nullptr,
                       (const int *)nullptr, (const float *)nullptr, (const float *)nullptr, make_sample_params(e, true));
    HIPCHK(hipGetLastError());
    return launch_post(e, MODE_OBS, ids, n, st, ids32, count_dev, /*prep_done=*/true, e->d_reset_calls);
}

extern "C" int parc_env_reset(ParcEnv *e, const int64_t *ids, int32_t k, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (k == 0) return PARC_OK;
    if (k > e->N) return fail(PARC_ERR_INVALID, "k > num_envs");
    if (k > 0 && !ids) return fail(PARC_ERR_INVALID, "env_ids is NULL");
    return sampling_reset(e, k < 0 ? nullptr : ids, nullptr, nullptr, k < 0 ? e->N : k, (hipStream_t)stream);
}

// Reset every env whose done flag was raised by the last step (base_agent.py:366-370) without the
// nonzero()/host round trip: the step kernel's compacted done list and its device-side count drive the launch.
// The list is consumed by the call: a second call before the next step resets nobody.
extern "C" int parc_env_reset_done(ParcEnv *e, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (!e->done_list_fresh) return PARC_OK;
    e->done_list_fresh = false;
    return sampling_reset(e, nullptr, e->d_done_list, e->d_reset_count + 1, e->N, (hipStream_t)stream);
}

// ---- whole control step as one hipGraph launch ----------------------------------------------------------------------
// step + reset_done are ~10 dependent launches; at a few thousand envs the host-side launch cost exceeds the kernels.
// The sequence is captured once on a private stream and replayed with one hipGraphLaunch.  Everything a kernel argument
// bakes in is either fixed per handle (buffers, tables), re-captured when it changes (graph_dirty), or read from device
// memory (done count, reset-call index, the bound action buffer).
extern "C" int parc_env_bind_action(ParcEnv *e, const float *action_dev) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    e->action_bound = action_dev;
    e->graph_dirty = true;
    return PARC_OK;
}

extern "C" int parc_env_step_reset_graph(ParcEnv *e, void *stream) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (e->cfg.enable_dynamics && !e->action_bound) return fail(PARC_ERR_STATE, "parc_env_bind_action must precede the graph step when dynamics is on");
    if (e->graph_dirty || !e->graph_exec) {
        if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
        hipStream_t cs = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (err == hipSuccess) {
            const bool timing = e->timing;
            e->timing = false; // events are not captured
            rc = parc_env_step(e, e->action_bound, cs);
            if (!rc) rc = parc_env_reset_done(e, cs);
            e->timing = timing;
            err = hipStreamEndCapture(cs, &graph);
        }
        if (err == hipSuccess && !rc) err = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipStreamDestroy(cs);
        if (rc) return rc;
        if (err != hipSuccess) { e->graph_exec = nullptr; return fail(PARC_ERR_HIP, std::string("graph capture failed: ") + hipGetErrorString(err)); }
        e->graph_dirty = false;
    }
    HIPCHK(hipGraphLaunch(e->graph_exec, (hipStream_t)stream));
    e->done_list_fresh = false; // the graph holds step + reset_done
    return PARC_OK;
}

extern "C" int parc_env_get_fail_rates(ParcEnv *e, float *out, int32_t M) {
    if (!e || !out || !e->have_motions || M != e->M) return fail(PARC_ERR_INVALID, "bad fail-rate query");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, e->d_fail, sizeof(float) * M, hipMemcpyDeviceToHost));
    return PARC_OK;
}

extern "C" int parc_env_set_fail_rates(ParcEnv *e, const float *in, int32_t M) {
    if (!e || !in || !e->have_motions || M != e->M) return fail(PARC_ERR_INVALID, "bad fail-rate update");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(e->d_fail, in, sizeof(float) * M, hipMemcpyHostToDevice));
    return PARC_OK;
}

extern "C" int parc_env_get_motion_info(ParcEnv *e, float *lengths, float *weights, int32_t M) {
    if (!e || !e->have_motions || M != e->M) return fail(PARC_ERR_INVALID, "bad motion-info query");
    for (int m = 0; m < M; ++m) {
        if (lengths) lengths[m] = e->h_meta[m].length;
        if (weights) weights[m] = e->h_weights[m];
    }
    return PARC_OK;
}

extern "C" int parc_env_set_never_done(ParcEnv *e, int32_t never_done) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    e->sp.never_done = never_done != 0;
    return sync_params(e);
}

extern "C" int parc_env_set_rand_reset(ParcEnv *e, int32_t rand_reset, int32_t demo_mode, float scale) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    e->cfg.rand_reset = rand_reset; e->cfg.demo_mode = demo_mode; e->cfg.rand_root_pos_offset_scale = scale;
    e->graph_dirty = true;
    return PARC_OK;
}

extern "C" int parc_env_set_start_time_fraction(ParcEnv *e, const float *frac_dev) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    e->d_start_frac = const_cast<float *>(frac_dev);
    e->graph_dirty = true;
    return PARC_OK;
}

extern "C" int parc_dof_to_rot(ParcEnv *e, const float *dof, float *jr, int32_t n, void *stream) {
    if (!e || !dof || !jr || n < 0) return fail(PARC_ERR_INVALID, "bad argument");
    if (n == 0) return PARC_OK;
    hipLaunchKernelGGL(k_dof_to_rot, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d_tab, dof, jr, n, e->B, e->D);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

extern "C" int parc_rot_to_dof(ParcEnv *e, const float *jr, float *dof, int32_t n, void *stream) {
    if (!e || !dof || !jr || n < 0) return fail(PARC_ERR_INVALID, "bad argument");
    if (n == 0) return PARC_OK;
    hipLaunchKernelGGL(k_rot_to_dof, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d_tab, jr, dof, n, e->B, e->D);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

extern "C" int parc_forward_kinematics(ParcEnv *e, const float *rp, const float *rr, const float *jr, float *bp, float *br, int32_t n,
                                       void *stream) {
    if (!e || !rp || !rr || !jr || !bp || n < 0) return fail(PARC_ERR_INVALID, "bad argument");
    if (n == 0) return PARC_OK;
    hipLaunchKernelGGL(k_fk, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d_tab, rp, rr, jr, bp, br, n, e->B);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

extern "C" int parc_calc_motion_frame(ParcEnv *e, const int32_t *ids, const float *times, int32_t n, float *rp, float *rr, float *rv,
                                      float *rav, float *jr, float *dv, float *ct, void *stream) {
    if (!e || !e->have_motions) return fail(PARC_ERR_STATE, "load_motions first");
    if (!ids || !times || !rp || !rr || !jr || n < 0) return fail(PARC_ERR_INVALID, "bad argument");
    if (n == 0) return PARC_OK;
    FrameOut F;
    F.root_pos = rp; F.root_rot = rr; F.root_vel = rv; F.root_ang_vel = rav; F.joint_rot = jr; F.dof_vel = dv; F.contacts = ct;
    hipLaunchKernelGGL(k_calc_motion_frame, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d_records, e->d_meta, ids, times, n,
                       e->B, e->J, e->D, F);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

// ---- test entry point: the device quaternion functions, element-wise ------------------------------------------------
__global__ void k_test_quat_op(int op, const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ t, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Q4 qa = (op == PARC_QOP_NORMALIZE3 || op == PARC_QOP_EXP_MAP_TO_QUAT || op == PARC_QOP_AA_TO_QUAT || op == PARC_QOP_ROTATE_2D)
                      ? mk4(a[3 * i], a[3 * i + 1], a[3 * i + 2], 0.f) : mk4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
    const V3 va = mk3(qa.x, qa.y, qa.z);
    Q4 qb = mk4(0.f, 0.f, 0.f, 1.f);
    V3 vb = mk3(0.f, 0.f, 0.f);
    if (b) {
        if (op == PARC_QOP_ROTATE) vb = mk3(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
        else qb = mk4(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
    }
    const float ts = t ? t[i] : 0.f;
    auto put4 = [&](Q4 q) { out[4 * i] = q.x; out[4 * i + 1] = q.y; out[4 * i + 2] = q.z; out[4 * i + 3] = q.w; };
    auto put3 = [&](V3 v) { out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; };
    switch (op) {
    case PARC_QOP_MUL: put4(quat_mul(qa, qb)); break;
    case PARC_QOP_ROTATE: put3(quat_rotate(qa, vb)); break;
    case PARC_QOP_CONJ: put4(quat_conj(qa)); break;
    case PARC_QOP_POS: put4(quat_pos(qa)); break;
    case PARC_QOP_NORMALIZE3: put3(normalize3(va)); break;
    case PARC_QOP_TO_AXIS_ANGLE: { V3 ax; float an; quat_to_axis_angle(qa, ax, an); put4(mk4(ax.x, ax.y, ax.z, an)); break; }
    case PARC_QOP_AA_TO_QUAT: put4(axis_angle_to_quat(va, ts)); break;
    case PARC_QOP_EXP_MAP_TO_QUAT: put4(exp_map_to_quat(va)); break;
    case PARC_QOP_TO_EXP_MAP: put3(quat_to_exp_map(qa)); break;
    case PARC_QOP_DIFF_ANGLE: out[i] = quat_diff_angle(qa, qb); break;
    case PARC_QOP_NORMALIZE: put4(quat_normalize(qa)); break;
    case PARC_QOP_TO_TAN_NORM: { float tn[6]; quat_to_tan_norm(qa, tn); for (int c = 0; c < 6; ++c) out[6 * i + c] = tn[c]; break; }
    case PARC_QOP_SLERP: put4(slerp(qa, qb, ts)); break;
    case PARC_QOP_HEADING: out[i] = calc_heading(qa); break;
    case PARC_QOP_HEADING_QUAT_INV: put4(heading_quat_inv(calc_heading(qa))); break;
    case PARC_QOP_DIFF: put4(quat_mul(qb, quat_conj(qa))); break; // torch_util.py:454 quat_diff(q0, q1) = q1 (x) conj(q0)
    case PARC_QOP_SLERP_RR: put4(slerp_rr(qa, qb, ts)); break; // the observation kernel's reduced-range variant
    case PARC_QOP_ROTATE_2D: { // torch_util.py:651, as the ray loop of k_env_post evaluates it
        const float ch = cosf(ts), sh = sinf(ts);
        out[2 * i] = qa.x * ch - qa.y * sh; out[2 * i + 1] = qa.x * sh + qa.y * ch; break;
    }
    default: break;
    }
}

extern "C" int parc_test_quat_op(int32_t op, const float *a, const float *b, const float *t, int32_t n, float *out, void *stream) {
    if (!a || !out || n < 0 || op < 0 || op > PARC_QOP_SLERP_RR) return fail(PARC_ERR_INVALID, "bad argument");
    if ((op == PARC_QOP_MUL || op == PARC_QOP_ROTATE || op == PARC_QOP_DIFF_ANGLE || op == PARC_QOP_SLERP || op == PARC_QOP_SLERP_RR || op == PARC_QOP_DIFF) && !b)
        return fail(PARC_ERR_INVALID, "this op needs a second operand");
    if ((op == PARC_QOP_AA_TO_QUAT || op == PARC_QOP_SLERP || op == PARC_QOP_SLERP_RR || op == PARC_QOP_ROTATE_2D) && !t) return fail(PARC_ERR_INVALID, "this op needs the scalar operand");
    if (n == 0) return PARC_OK;
    hipLaunchKernelGGL(k_test_quat_op, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, (int)op, a, b, t, (int)n, out);
    HIPCHK(hipGetLastError());
    return PARC_OK;
}

#ifndef PARC_BUILD_FLAGS
#define PARC_BUILD_FLAGS "unknown"
#endif
extern "C" const char *parc_build_flags(void) { return PARC_BUILD_FLAGS; }

extern "C" int parc_env_get_frame_vel_tables(ParcEnv *e, float *root_vel, float *root_ang_vel, float *dof_vel) {
    if (!e || !e->have_motions) return fail(PARC_ERR_STATE, "load_motions first");
    HIPCHK(hipSetDevice(e->cfg.device));
    std::vector<float> rec((size_t)e->F * 128);
    HIPCHK(hipMemcpy(rec.data(), e->d_records, rec.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t f = 0; f < e->F; ++f) {
        const float *v = rec.data() + (size_t)f * 128 + 4 * REC_Q_VEL;
        if (root_vel) for (int c = 0; c < 3; ++c) root_vel[3 * f + c] = v[c];
        if (root_ang_vel) for (int c = 0; c < 3; ++c) root_ang_vel[3 * f + c] = v[4 + c];
        if (dof_vel) for (int d = 0; d < e->D; ++d) dof_vel[(size_t)f * e->D + d] = v[8 + d];
    }
    return PARC_OK;
}

extern "C" const unsigned int *parc_env_health_words(ParcEnv *e) { return (e && e->d_health) ? e->h_health : nullptr; }

extern "C" const char *parc_env_describe(ParcEnv *e) {
    if (!e) return "";
    char b[1024];
    std::string d;
    const bool ctrl_ff = e->cfg.enable_dynamics && e->cfg.dynamics.control_mode != PARC_CTRL_PD;
    const char *dk = !e->cfg.enable_dynamics ? "none" : (e->use_wave ? (ctrl_ff ? "k_dynamics_wave_ff" : "k_dynamics_wave") : (e->use_coop ? "k_dynamics_coop" : "k_dynamics"));
    snprintf(b, sizeof(b), "dynamics_kernel=%s;", dk); d += b;
    if (e->cfg.enable_dynamics) {
        const parcdyn::DynModel &m = e->h_dyn;
        static const char *const ctrl_names[] = {"pd", "vel", "torque", "pd_exp", "pd_1d"};
        snprintf(b, sizeof(b), "control_mode=%s;", ctrl_names[m.ctrl]); d += b;
        snprintf(b, sizeof(b), "envs_per_block=%d;substeps=%d;substep_dt=%.9g;collision_points=%d;collision_segments=%d;kn=%g;dn=%g;dtang=%g;mu=%g;pen_cap=%g;"
                 "manifold_period=%d;spec_m0=%g;spec_tv=%g;spec_max=%g;root_residual=%d;", e->use_wave ? wave_envs_per_block(e->N, e->num_cus) : (e->use_coop ? CO_ENVS : 64),
                 m.nsub, (double)m.dt, m.ncol, m.nseg, (double)m.kn, (double)m.dn, (double)m.dtang, (double)m.mu, (double)m.pen_cap, m.man_period, (double)m.spec_m0,
                 (double)m.spec_tv, (double)m.spec_max, e->d_root_shadow ? 1 : 0);
        d += b;
        if (e->wave_rejected) d += "note=the model does not fit k_dynamics_wave (tree shape or > 32 candidates on a body): general fallback kernel;";
        if (e->use_wave) { snprintf(b, sizeof(b), "manifold_lds_slots=%d+%d+%d+%d;manifold_overflow_slots=%d;", e->h_wave.man_cap[0], e->h_wave.man_cap[1], e->h_wave.man_cap[2], e->h_wave.man_cap[3], WV_MAN_OVF); d += b; }
    }
    const bool small = e->M <= 64 && !e->force_ema_leader && e->N <= CURRICULUM_ONE_LAUNCH_MAX && (e->N & 7) == 0 && !e->force_two_launch_curriculum;
    snprintf(b, sizeof(b), "curriculum=%s;post_kernel=%s;dev_options=%s", small ? "k_curriculum_small" : ((e->M <= 64 && !e->force_ema_leader) ? "k_done_scatter+k_fail_rate_ema" : "k_done_scatter+k_ema_first+k_ema_leader"),
             e->bound ? parc_env_post_kernel(e) : "unbound", e->dev_options_s.empty() ? "none" : e->dev_options_s.c_str());
    d += b;
    e->describe_s = d;
    return e->describe_s.c_str();
}

extern "C" int parc_env_dynamics_timeouts(ParcEnv *e) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    unsigned int v = 0;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(parcdyn::g_wave_timeouts), sizeof(v)));
    return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}

// Contact planes k_dynamics_wave had no room for (the per-lane list's LDS share + overflow area were full; parc_dynamics_wave.hpp,
// WvMan): must stay 0 -- a dropped plane is a contact the substeps after a discovery do not see.
extern "C" int parc_env_dynamics_manifold_drops(ParcEnv *e) {
    if (!e) return fail(PARC_ERR_INVALID, "null env");
    unsigned int v = 0;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(parcdyn::g_wave_man_drops), sizeof(v)));
    return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}

extern "C" float parc_env_last_dynamics_ms(ParcEnv *e) { return e ? e->last_dyn_ms : 0.f; }

extern "C" const char *parc_env_dynamics_kernel(ParcEnv *e) {
    if (!e || !e->cfg.enable_dynamics) return "";
    return e->use_wave ? "k_dynamics_wave" : (e->use_coop ? "k_dynamics_coop" : "k_dynamics");
}

extern "C" const char *parc_env_post_kernel(ParcEnv *e) {
    if (!e || !e->bound) return "";
    return (wants_mirror(e->sp.buf) || obs_variant(e->cfg)) ? "k_env_post<MODE,true>" : "k_env_post<MODE,false>";
}

#ifdef PARC_STAMPS
// Diagnostic (-DPARC_STAMPS builds only; the shipped library does not export it): mean cycles per phase of k_env_post
// over all envs of the last step.
extern "C" int parc_env_debug_stamps(ParcEnv *e, double *mean8) {
    if (!e || !e->sp.stamp_out) return fail(PARC_ERR_STATE, "no stamp buffer");
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)e->N * 8);
    HIPCHK(hipMemcpy(h.data(), e->sp.stamp_out, h.size() * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) { double a = 0; for (int i = 0; i < e->N; ++i) a += h[(size_t)i * 8 + k]; mean8[k] = a / e->N; }
    return PARC_OK;
}

// Diagnostic: cycles per phase of k_dynamics_coop summed over all waves since the last call (then cleared).
extern "C" int parc_env_debug_dyn_stamps(double *out16) {
    unsigned long long h[16], z[16] = {0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(parcdyn::g_dyn_stamps), sizeof(h)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(parcdyn::g_dyn_stamps), z, sizeof(z)));
    for (int i = 0; i < 16; ++i) out16[i] = (double)h[i];
    return PARC_OK;
}

// Diagnostic: cycles per segment of k_dynamics_wave, per wave role [4][16], summed over all blocks since the last call.
extern "C" int parc_env_debug_wave_stamps(double *out64) {
    unsigned long long h[64], z[64] = {0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(parcdyn::g_wave_stamps), sizeof(h)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(parcdyn::g_wave_stamps), z, sizeof(z)));
    for (int i = 0; i < 64; ++i) out64[i] = (double)h[i];
    return PARC_OK;
}

// Diagnostic: contact-cull statistics of k_dynamics_wave per body [16][8] (see g_wave_cnt), cleared by the call.
extern "C" int parc_env_debug_wave_counts(double *out128) {
    unsigned long long h[128], z[128] = {0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(parcdyn::g_wave_cnt), sizeof(h)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(parcdyn::g_wave_cnt), z, sizeof(z)));
    for (int i = 0; i < 128; ++i) out128[i] = (double)h[i];
    return PARC_OK;
}
#endif

#ifdef PARC_STAMPS
// Diagnostic: manifold-size histograms of k_dynamics_wave [16][4][16] (see g_wave_hist; -DPARC_COUNTS builds fill them), cleared by the call.
extern "C" int parc_env_debug_wave_hist(double *out1024) {
    static unsigned long long h[1024], z[1024];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(parcdyn::g_wave_hist), sizeof(h)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(parcdyn::g_wave_hist), z, sizeof(z)));
    for (int i = 0; i < 1024; ++i) out1024[i] = (double)h[i];
    return PARC_OK;
}
#endif

#ifdef PARC_TIMELINE
// Diagnostic (-DPARC_TIMELINE builds): absolute cycle counter at the hand-off points of one substep of block 0, [wave][event]
extern "C" int parc_env_debug_wave_timeline(double *out64) {
    unsigned long long h[64];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(parcdyn::g_wave_tl), sizeof(h)));
    for (int i = 0; i < 64; ++i) out64[i] = (double)h[i];
    return PARC_OK;
}
#endif

extern "C" int parc_env_profile_step(ParcEnv *e, const float *action_dev, void *stream, int32_t iters, float *avg_ms, float *avg_post_ms) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (iters < 1) return fail(PARC_ERR_INVALID, "iters must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    double tot = 0.0, post = 0.0, dyn = 0.0;
    for (int i = 0; i < iters; ++i) {
        HIPCHK(hipEventRecord(e->ev[3], st));
        rc = launch_dynamics(e, action_dev, st);
        if (rc) return rc;
        HIPCHK(hipEventRecord(e->ev[0], st));
        rc = launch_post(e, MODE_STEP, nullptr, e->N, st);
        if (rc) return rc;
        HIPCHK(hipEventRecord(e->ev[1], st));
        rc = launch_curriculum(e, st);
        if (rc) return rc;
        HIPCHK(hipEventRecord(e->ev[2], st));
        HIPCHK(hipEventSynchronize(e->ev[2]));
        float a = 0.f, b = 0.f, c = 0.f;
        HIPCHK(hipEventElapsedTime(&a, e->ev[3], e->ev[2]));
        HIPCHK(hipEventElapsedTime(&b, e->ev[0], e->ev[1]));
        HIPCHK(hipEventElapsedTime(&c, e->ev[3], e->ev[0]));
        tot += a; post += b; dyn += c;
    }
    if (avg_ms) *avg_ms = (float)(tot / iters);
    if (avg_post_ms) *avg_post_ms = (float)(post / iters);
    e->last_dyn_ms = (float)(dyn / iters);
    return PARC_OK;
}
