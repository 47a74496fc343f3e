#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the motion-tracking env step at 65 536 envs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            # launches its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # or runs as one of N externally launched ranks

A "step" is what the PPO rollout does to the env per control step (base_agent.py:348-370): ``env.step(action)``
followed by the reset of the envs that finished.  Envs shard across ranks as contiguous global-index ranges with no
data-path collective.  ``--scaling strong`` (default, the north-star metric): ``--envs`` is the TOTAL (65 536 ->
8 192 per GPU at 8 GPUs); ``--scaling weak``: ``--envs`` per GPU.  Rank 0 prints ONE JSON line.

``--ppo 1`` adds the learner's collective to every 32 steps (cfg 4): 40 all-reduces of the 42.56 MB fp32 gradient bucket
over RCCL, i.e. env-steps/s of a PPO iteration's env + communication legs (the policy network itself is PyTorch's and is
not part of this benchmark).

The JSON carries ``roofline`` (dominant kernel; hipEvent-measured duration on the launch stream) and, at N=1,
``cpu_baseline`` (the CPU oracle — a scalar C port of the reference step, pinned against the reference's golden vectors —
timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# SURVEY.md §8(d) / BASELINE.md §4: algorithmic HBM bytes per env-step
BYTES_KINEMATIC = 5772   # read state 276 + contact forces 180 + bookkeeping 24; write obs 5248 + reward/done/terms 44
BYTES_DYNAMICS = 6340    # + action 112, state write-back 276, contact-force write 180
BYTES_DYN_KERNEL = 844   # k_dynamics alone: state 276 + action 112 in, state 276 + contact forces 180 out
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak = 1024 SIMDs x 64 lanes x 2 flop / 2 cycles x 2.4 GHz
GRAD_BUCKET_FLOATS = 10638877  # parameters of the default actor/critic (dm_agent_default.yaml), one flat fp32 bucket = 42.56 MB
PPO_STEPS_PER_ITER, PPO_ALLREDUCES_PER_ITER = 32, 40  # dm_agent_default.yaml: steps_per_iter, update_epochs x minibatches


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="total envs (--scaling strong) or envs per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--config", default=os.path.join(REPO, "data/configs/tracker_config/dm_env_default.yaml"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--motions", type=int, default=0,
                    help="synthetic library of this many entries from the bundled clips (SURVEY 8(d): cfg 3 = 1024, cfg 4/5 = 16384); 0 = the clips as they are")
    ap.add_argument("--yaw", type=int, default=0, help="with --motions: 1 = pseudo-clips yaw-rotated by 2 pi i / M, weights = clip length (cfg 4/5)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: step + reset_done as one hipGraph launch (parc_env_step_reset_graph); kernel timings are then taken from a short separate run")
    ap.add_argument("--dynamics", type=int, default=1, help="1: full step (rigid-body dynamics + contact), 0: kinematic step only")
    ap.add_argument("--ppo", type=int, default=0, help="1: add the 40 x 42.56 MB gradient all-reduces of a PPO iteration to every 32 steps (cfg 4)")
    ap.add_argument("--kernel-sample-steps", type=int, default=200,
                    help="UNTIMED steps after the timed region whose per-step hipEvent kernel durations give roofline.kernel_sample (median / p95): "
                         "a thicker sample than a short --steps; `value` and ms_per_step never include them")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: the parent never touches the GPU; it starts one fresh process per rank and relays rank 0's JSON line
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, script=None, extra_env=None, deadline_s=None, poll_s=0.05):
    """Start ``n`` fresh ranks of ``script`` (default: this file) with torchrun-style env vars and SUPERVISE them:
    all children are polled; the first non-zero exit (or the overall deadline, rc 124) terminates, then kills, the
    siblings — a rank that dies after the rendezvous must not leave the others inside a collective until the
    NCCL / gloo timeout.  Rank 0's stdout is drained by a thread.  Returns (rc, rank-0 stdout)."""
    port = _free_port()
    if deadline_s is None:
        deadline_s = float(os.environ.get("PARC_BENCH_DEADLINE_S", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    drain.start()
    t_end = time.monotonic() + deadline_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > t_end:
            rc = 124
            sys.stderr.write(f"bench.py: ranks still running after {deadline_s:.0f} s, stopping them\n")
            break
        time.sleep(poll_s)
    if rc != 0:  # stop the survivors: exactly the PIDs started above
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    drain.join(timeout=10.0)
    return rc, (chunks[0].decode() if chunks else "")


def _count_gpus():
    """GPUs visible to this user WITHOUT a HIP / HSA call in the launcher process: KFD topology nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are set.  Falls back to torch's count when sysfs is not there."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(ln.split()[:2] for ln in open(f).read().splitlines() if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            pass
    if n == 0:
        import torch
        return torch.cuda.device_count()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def shard_sizes(envs, world, scaling):
    """(envs per GPU, total envs): strong = ``envs`` is the total, split evenly; weak = ``envs`` per GPU."""
    if scaling == "strong":
        if envs % world != 0:
            raise ValueError("--envs must be divisible by --gpus for strong scaling")
        return envs // world, envs
    return envs, envs * world


def _cpu_model():
    """Model name of the host CPU (core count and CPU model are stated with the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(env, seconds, dynamics_on, actions):
    """Time the CPU oracle (oracle/parc_oracle.c [+ the host build of the dynamics core]) on the same scene/state,
    all host cores, bounded sample."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.binding import Oracle
    import helpers
    from conftest import golden
    oracle = Oracle()
    cg = golden("char_model")
    oc = oracle.make_char(cg["parent"], cg["local_translation"], cg["local_rotation"], cg["joint_type"], cg["joint_axis"],
                          cg["dof_idx"], int(cg["dof_size"]))
    sc = env._scene
    n = env.get_num_envs()
    clips = [dict(name=c.name, root_pos=c.root_pos, root_rot=c.root_rot, joint_rot=c.joint_rot, contacts=c.contacts, fps=c.fps,
                  loop_mode=c.loop_mode) for c in sc.clips]
    lib = oracle.mlib_create(oc, clips, [c.weight for c in sc.clips])
    ocfg = helpers.default_cfg(oracle, n, sc.ray_points, sc.env_offsets, sc.grid.motion_offsets)
    ter = oracle.make_terrain(sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy)
    st = oracle.make_state(n, M=len(clips), tracking_error=False)
    for ko, ke in [("char_root_pos", "_char_root_pos"), ("char_root_rot", "_char_root_rot"), ("char_root_vel", "_char_root_vel"),
                   ("char_root_ang_vel", "_char_root_ang_vel"), ("char_dof_pos", "_char_dof_pos"), ("char_dof_vel", "_char_dof_vel"),
                   ("contact_forces", "_char_contact_forces"), ("time_offsets", "_motion_time_offsets"),
                   ("char_body_pos", "_char_rigid_body_pos")]:
        st[ko][...] = getattr(env, ke).cpu().numpy()
    st["motion_ids"][...] = env._motion_ids.cpu().numpy(); st["terrain_ids"][...] = env._motion_terrain_ids.cpu().numpy()
    cores = max(1, min(os.cpu_count() or 1, 64))
    bounds = np.linspace(0, n, cores + 1).astype(int)

    dyn = None
    if dynamics_on:
        from oracle.binding_dyn import DynOracle
        dyn = DynOracle(sc.cfg)
        act = actions[0].cpu().numpy()
        hf_t, mp_t, dx_t = sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy
        dst = dict(root_pos=st["char_root_pos"], root_rot=st["char_root_rot"], root_vel=st["char_root_vel"],
                   root_ang_vel=st["char_root_ang_vel"], dof_pos=st["char_dof_pos"], dof_vel=st["char_dof_vel"],
                   contact_force=st["contact_forces"])

    def work(b0, b1):
        if dyn is not None:
            dyn.step(hf_t, mp_t, dx_t, {k: v[b0:b1] for k, v in dst.items()}, act[b0:b1], sc.env_offsets[b0:b1])
            jr = oracle.dof_to_rot(oc, st["char_dof_pos"][b0:b1])
            st["char_body_pos"][b0:b1] = oracle.forward_kinematics(oc, st["char_root_pos"][b0:b1], st["char_root_rot"][b0:b1], jr)[0]
        oracle.env_post_physics_step(oc, lib, ter, ocfg, st, b0, b1)

    def run_step():
        th = [threading.Thread(target=work, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(cores)]
        for t in th: t.start()
        for t in th: t.join()
        oracle.env_update_curriculum(lib, ocfg, st)

    run_step()  # warm-up
    st["timestep_buf"][:] = 0
    t0 = time.time(); steps = 0
    while steps < 3 or (time.time() - t0 < seconds and steps < 10000):
        st["timestep_buf"][:] = steps % 8  # stay inside the clips
        run_step(); steps += 1
    dt = time.time() - t0
    out = {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "port",
           "sample": f"{steps} {'full (dynamics + obs/reward/done)' if dynamics_on else 'kinematic'} steps x {n} envs, C/C++ oracle "
                     "(scalar, -O2, one thread per core), same scene and state"}
    try:
        out["torch_path"] = torch_path_baseline(env, max(3.0, 0.5 * seconds), st, oracle, lib)
    except Exception as ex:  # a baseline leg must never take the benchmark down
        out["torch_path"] = {"error": repr(ex)}
    return out


def torch_path_baseline(env, seconds, st, oracle, orc_lib):
    """The north star's "reference CPU-PyTorch motion_lib / kin_char_model path": oracle/torch_path.py issues the reference's batched
    tensor-op sequence (calc_motion_frame at t and the six look-ahead times, dof <-> rot, the three FK calls of a step) on CPU
    tensors with all host cores.  Pinned against the reference's golden vectors by tests/test_oracle_golden.py."""
    import numpy as np
    import torch
    from oracle import torch_path as tp
    from conftest import golden
    cg = golden("char_model")
    cm = tp.CharModel(cg["parent"], cg["local_translation"], cg["local_rotation"], cg["joint_type"], cg["joint_axis"], cg["dof_idx"], int(cg["dof_size"]))
    sc = env._scene
    clips = sc.clips
    F = sum(c.num_frames for c in clips)
    # load-time velocity tables of the CPU oracle's own motion library (MotionLib._load_motion_file :318-327), not the product's
    rv, rav, dv = (oracle.mlib_array(orc_lib, "frame_root_vel", (F, 3)), oracle.mlib_array(orc_lib, "frame_root_ang_vel", (F, 3)),
                   oracle.mlib_array(orc_lib, "frame_dof_vel", (F, 28)))
    nf = np.array([c.num_frames for c in clips], np.int64)
    tables = tp.make_tables([dict(root_pos=c.root_pos, root_rot=c.root_rot, joint_rot=c.joint_rot, contacts=c.contacts, fps=c.fps, loop_mode=c.loop_mode) for c in clips],
                            rv, rav, dv)
    lib = tp.MotionLib(tables)
    cores = max(1, min(os.cpu_count() or 1, 64))
    torch.set_num_threads(cores)
    n = env.get_num_envs()
    ids = torch.as_tensor(st["motion_ids"].astype(np.int64))
    times = torch.as_tensor(st["time_offsets"].astype(np.float32))
    crp, crr, cdof = torch.as_tensor(st["char_root_pos"].copy()), torch.as_tensor(st["char_root_rot"].copy()), torch.as_tensor(st["char_dof_pos"].copy())
    with torch.no_grad():
        tp.step_path(cm, lib, ids, times, crp, crr, cdof, 1.0 / 30.0)  # warm-up
        t0 = time.time(); steps = 0
        while steps < 2 or (time.time() - t0 < seconds and steps < 1000):
            tp.step_path(cm, lib, ids, times + steps / 30.0, crp, crr, cdof, 1.0 / 30.0); steps += 1
        dt = time.time() - t0
    out = {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{steps} x {n} envs: PyTorch-CPU op sequence of the reference's motion_lib.calc_motion_frame (7 samples per env) + kin_char_model "
                     "dof_to_rot / rot_to_dof / forward_kinematics (character, reference, 6 targets), torch.set_num_threads(cores); no terrain rays, "
                     "observation assembly or reward"}
    # the WHOLE kinematic step as the reference issues it (oracle/torch_path.py::KinematicStep: + height rays, observation, reward, done;
    # pinned against the reference's own _post_physics_step outputs by tests/test_oracle_golden.py): BASELINE.md section 2's measurement
    # (30.9 k env-steps/s for the reference itself on 8 cores at 65 536 envs), repeated on this box's cores
    c = sc.cfg
    B = c.model.num_bodies
    ks = tp.KinematicStep(cm, lib, tp.Terrain(sc.grid.terrain.hf, sc.grid.terrain.min_point, sc.grid.terrain.dxdy),
                          dict(key_body_ids=sc.key_body_ids, tar_obs_steps=[c.tar_obs_steps[i] for i in range(c.num_tar_obs_steps)], ray_points=sc.ray_points,
                               env_offsets=sc.env_offsets, motion_offsets=sc.grid.motion_offsets, timestep=c.control_dt, episode_length=c.episode_length,
                               min_obs_h=c.min_obs_h, max_obs_h=c.max_obs_h, reward_weights=[c.pose_w, c.vel_w, c.root_pos_w, c.root_vel_w, c.key_pos_w],
                               joint_err_w=sc.joint_err_w, dof_err_w=sc.dof_err_w, contact_weights=[c.contact_weights[b] for b in range(B)],
                               pose_termination_dist=[c.pose_termination_dist[j] for j in range(B - 1)],
                               root_pos_termination_dist=c.root_pos_termination_dist, root_rot_termination_angle=c.root_rot_termination_angle))
    fs = {k: torch.as_tensor(st[k].copy()) for k in ("char_root_pos", "char_root_rot", "char_root_vel", "char_root_ang_vel", "char_dof_pos", "char_dof_vel",
                                                     "char_body_pos", "contact_forces", "time_offsets")}
    fs["motion_ids"] = ids; fs["terrain_ids"] = torch.as_tensor(st["terrain_ids"].astype(np.int64))
    with torch.no_grad():
        fs["timestep"] = torch.zeros(n, dtype=torch.int32)
        ks.step(fs)  # warm-up
        t0 = time.time(); fsteps = 0
        while fsteps < 2 or (time.time() - t0 < seconds and fsteps < 1000):
            fs["timestep"] = torch.full((n,), fsteps % 8, dtype=torch.int32)   # stay inside the clips
            ks.step(fs); fsteps += 1
        fdt = time.time() - t0
    out["full_step"] = {"value": n * fsteps / fdt, "unit": "env-steps/s", "cores": cores, "kind": "port",
                        "sample": f"{fsteps} x {n} envs: PyTorch-CPU op sequence of the reference's whole kinematic step (441-ray height gather, reference frame + FK, 6 targets, "
                                  "1 312-column observation, DeepMimic reward + contact term, compute_done), torch.set_num_threads(cores); no physics, no fail-rate EMA loop"}
    return out


def _profile_json(suffix, build=None):
    """Newest profiles/rNN_<suffix>.  With ``build`` = (csrc hash, build flags): only a file collected on exactly that build of the
    library (tools/profile_summary.py stores both) -- counter figures of another build would silently describe a different kernel."""
    import glob
    for p in sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
        d = json.load(open(p))
        if build is None or (d.get("csrc_sha16") == build[0] and d.get("build_flags") == build[1]):
            d["_file"] = os.path.relpath(p, REPO)
            return d
    return None


def worker(a):
    import ctypes as C
    import torch
    from parc_amd import lib as L
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # rehearsal switch for a one-GPU box (never set by the driver): all ranks on cuda:0, gloo for the barrier / max
    share = bool(os.environ.get("PARC_BENCH_SHARE_GPU"))
    backend = os.environ.get("PARC_BENCH_BACKEND", "gloo" if share else "nccl")
    if share:
        local = 0
    dist = None
    if world > 1 or a.ppo:
        # --ppo at one rank still forms a (1-rank) process group and issues the all-reduces, so the RCCL path is loaded,
        # executed and timed on a single GPU as well (cfg 4's collective leg)
        import torch.distributed as dist
        kw = {}
        if "RANK" not in os.environ:
            kw = dict(init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"), **kw)
        else:
            dist.init_process_group(backend, **kw)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.util import path_loader
    cfg = path_loader.load_config(a.config)
    if a.motions > 0:  # synthetic library generated in memory (parc_amd/util/synth_dataset.py)
        import tempfile
        from parc_amd.util import synth_dataset
        d = tempfile.mkdtemp(prefix=f"parc_bench_lib_r{rank}_")
        base = str(path_loader.resolve_path(cfg["env"]["dm"]["motion_file"]))
        cfg["env"]["dm"]["motion_file"] = synth_dataset.write_spec(os.path.join(d, "motions.yaml"), base, a.motions, yaw=bool(a.yaw))
        cfg["env"]["dm"].pop("terrain_save_path", None)
    n_local, n_total = shard_sizes(a.envs, world, a.scaling)
    dyn = bool(a.dynamics)
    sys.stdout = sys.stderr if rank == 0 else open(os.devnull, "w")  # the only stdout line is the JSON below
    env = HipParkourEnv(cfg, n_local, dev, False, env_id_base=rank * n_local, total_envs=n_total, seed=1234 + rank,
                        mirror_ref_state=False, enable_dynamics=dyn)
    dynamics_on = bool(env._scene.cfg.enable_dynamics)
    mode = cfg["env"]["dm"].get("terrain_build_mode", "square")
    ncl = len(env._scene.clips)
    lib_desc = ((f"{a.motions} synthetic library entries (bundled clips " + ("yaw-rotated, weight = length)" if a.yaw else "replicated)")) if a.motions > 0 else
                (f"{ncl} bundled clips" if ncl > 1 else f"clip {env._scene.clips[0].name}")) + \
               {"square": " on a square blocky grid", "wide": " on a wide blocky grid", "file": " on its own terrain"}.get(mode, "")
    D = env._char_dof_pos.shape[1]
    # untrained-policy actions (SURVEY §8(d) cfg 3): action-normalizer mean + N(0, 0.05^2) * std
    lo, hi = env._action_bound_low, env._action_bound_high
    mean, std = 0.5 * (hi + lo), 0.5 * (hi - lo)
    actions = [mean + 0.05 * std * torch.randn(n_local, D, device=dev) for _ in range(4)]
    env.reset()
    grad = None
    if a.ppo:
        grad = torch.zeros(GRAD_BUCKET_FLOATS, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")

    coll_events = []

    def one_step(i):
        if a.graph:
            env.step_and_reset_done(actions[i & 3] if dynamics_on else None)
        else:
            env.step(actions[i & 3])
            env.reset_done()
        if grad is not None and (i + 1) % PPO_STEPS_PER_ITER == 0:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if grad.is_cuda else None
            if ev:
                ev[0].record()
            for _ in range(PPO_ALLREDUCES_PER_ITER):  # the learner's gradient bucket, once per minibatch (optimizer.py)
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                grad.mul_(1.0 / world)
            if ev:
                ev[1].record()
                coll_events.append(ev)

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    coll_events.clear()
    L.check(env._lib.parc_env_set_kernel_timing(env._handle, 1))  # hipEvents around the kernels of the timed steps, no sync
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # kernel durations over the timed region: hipEvents recorded on the launch stream by every parc_env_step
    import numpy as np

    def read_samples():
        cap = 65536
        bufs = [np.zeros(cap, np.float32) for _ in range(3)]
        got = C.c_int32()
        L.check(env._lib.parc_env_get_kernel_timing_samples(env._handle, *[b.ctypes.data_as(L.f32p) for b in bufs], cap, C.byref(got)))
        k = min(int(got.value), cap)
        return [b[:k].astype(np.float64) for b in bufs]

    def stats(x):
        return None if len(x) == 0 else {"n": int(len(x)), "mean": float(np.mean(x)), "median": float(np.median(x)), "p95": float(np.quantile(x, 0.95)),
                                         "min": float(np.min(x)), "max": float(np.max(x))}
    dms, pms, nst = C.c_double(), C.c_double(), C.c_int32()
    timed_samples = None if a.graph else read_samples()
    L.check(env._lib.parc_env_get_kernel_timing(env._handle, C.byref(dms), C.byref(pms), C.byref(nst)))
    L.check(env._lib.parc_env_set_kernel_timing(env._handle, 0))
    if a.graph:  # events are not part of the captured graph: time the kernels over 20 ordinary steps instead
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 1))
        for i in range(20):
            env.step(actions[i & 3]); env.reset_done()
        timed_samples = read_samples()
        L.check(env._lib.parc_env_get_kernel_timing(env._handle, C.byref(dms), C.byref(pms), C.byref(nst)))
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 0))
    else:
        assert nst.value == a.steps
    # a thicker, UNTIMED sample of the same kernels (the driver's --steps 20 is 15 ms of GPU time): per-step medians and p95
    extra_samples = None
    if a.kernel_sample_steps > 0:
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 1))
        for i in range(a.kernel_sample_steps):
            env.step(actions[i & 3]); env.reset_done()
        torch.cuda.synchronize()
        extra_samples = read_samples()
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 0))
    dyn_ms, post_ms = float(dms.value), float(pms.value)

    # Counter-derived figures come from the committed rocprofv3 --pmc passes (profiles/, collected on the 65 536-env
    # default configuration in separate runs), NOT from this run; they are attached only to that configuration.
    at_profiled_cfg = n_local == 65536 and a.motions == 0
    build = (L.csrc_hash(), env._lib.parc_build_flags().decode())
    post_traffic = None
    pmc = _profile_json("pmc_hbm_traffic.json", build)
    if pmc and at_profiled_cfg:
        k = [q for q in pmc["FETCH_SIZE_KiB_avg_per_dispatch"] if q.startswith("void k_env_post<0")][0]  # the step instantiation
        post_traffic = (2.0 * pmc["FETCH_SIZE_KiB_avg_per_dispatch"][k] + pmc["WRITE_SIZE_KiB_avg_per_dispatch"][k]) * 1024.0
    post_gbs = BYTES_KINEMATIC * n_local / (post_ms * 1e-3) / 1e9
    obs_kernel = {"kernel": "k_env_post<MODE_STEP> (+ k_env_prep when the dynamics kernel did not write the prep records)", "bound": "hbm", "kernel_ms": post_ms, "achieved": post_gbs, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": post_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": BYTES_KINEMATIC,
                  "traffic": post_traffic,
                  "traffic_source": ((pmc["_file"] + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes on this build of the library), not measured in this run")
                                     if post_traffic else "no counter file of this build / configuration under profiles/")}
    if dynamics_on and dyn_ms >= post_ms:
        # dominant kernel = the dynamics: VALU-issue bound (rigid-body recursion in registers), 844 B of HBM per env-step.
        kname = env._lib.parc_env_dynamics_kernel(env._handle).decode()
        hbm_gbs = BYTES_DYN_KERNEL * n_local / (dyn_ms * 1e-3) / 1e9
        dj = _profile_json("pmc_dynamics.json", build)   # only counters collected on THIS build (source hash + flags)
        valu = traffic = lane_util = None
        if dj and dj.get("kernel") == kname and at_profiled_cfg:
            traffic = 2.0 * dj["FETCH_SIZE_KiB"] * 1024.0 + dj["WRITE_SIZE_KiB"] * 1024.0
            valu = dj.get("SQ_INSTS_VALU")  # wave-level VALU instructions per launch
            lane_util = dj.get("derived", {}).get("lane_utilisation_of_valu")
        roof = {"bound": "valu", "kernel": kname, "kernel_ms": dyn_ms, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s-equivalent of VALU issue slots",
                "achieved": None, "frac": None, "traffic": traffic,
                "traffic_source": (dj["_file"] + " (separate rocprofv3 --pmc passes on this build of the library), not measured in this run") if valu else None,
                "definition": "VALU ISSUE-SLOT UTILISATION, not useful flops: achieved = wave-level VALU instructions per launch (SQ_INSTS_VALU, profiles/) x 128 "
                              "(64 lanes x 2) / measured kernel time, against the packed-fp32 vector peak; every VALU instruction counts (moves, compares, selects) "
                              "and idle lanes count as busy -- `lane_utilisation` and `frac_of_scalar_fp32_issue_rate` put it in proportion",
                "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_env_step": BYTES_DYN_KERNEL},
                "obs_kernel": obs_kernel}
        if valu:
            roof["achieved"] = valu * 128.0 / (dyn_ms * 1e-3) / 1e12
            roof["frac"] = roof["achieved"] / VALU_PEAK_TFLOPS
            roof["lane_utilisation"] = lane_util
            # a non-packed fp32 VALU instruction occupies its SIMD for 4 cycles: 1024 SIMDs x 2.4 GHz / 4 instructions per second
            roof["frac_of_scalar_fp32_issue_rate"] = valu / (dyn_ms * 1e-3) / (1024 * 2.4e9 / 4.0)
        else:  # no counter file for this configuration: fall back to the HBM figure the contract asks for
            roof.update({"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                         "note": "VALU-bound kernel; no SQ counter file for this configuration, HBM fraction reported instead"})
    else:
        roof = dict(obs_kernel)
        if dynamics_on:
            hbm_gbs = BYTES_DYN_KERNEL * n_local / (dyn_ms * 1e-3) / 1e9
            roof["dynamics_kernel"] = {"kernel": env._lib.parc_env_dynamics_kernel(env._handle).decode(), "kernel_ms": dyn_ms, "bound": "valu",
                                       "hbm": {"achieved": hbm_gbs, "frac": hbm_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": BYTES_DYN_KERNEL}}
    whole_bytes = BYTES_DYNAMICS if dynamics_on else BYTES_KINEMATIC
    roof["whole_step_algorithmic_bytes"] = whole_bytes
    # SURVEY 8(d)'s figure for the PATH: algorithmic bytes of the whole step / wall time of the timed region, against the HBM peak
    roof["hbm_whole_step_gbs"] = whole_bytes * n_total / (dt / a.steps) / 1e9 / world
    roof["hbm_whole_step_frac"] = roof["hbm_whole_step_gbs"] / HBM_PEAK_GBS
    names = ("dynamics_kernel_ms", "obs_kernel_ms", "curriculum_launches_ms")
    roof["kernel_ms_timed_steps"] = {nm: stats(x) for nm, x in zip(names, timed_samples)}
    if extra_samples is not None:
        roof["kernel_sample"] = {"steps": a.kernel_sample_steps, "note": "untimed steps after the timed region, hipEvents on the launch stream, per step",
                                 **{nm: stats(x) for nm, x in zip(names, extra_samples)}}
    out = {
        "metric": "env-steps/s", "value": n_total * a.steps / dt, "unit": "env-steps/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("full step (dynamics + obs/reward/done)" if dynamics_on else "kinematic step (ref slerp + FK + 441-ray hf + obs + reward + done), no physics")
                               + f", {n_total} envs total = {n_local} per GPU, " + lib_desc + ", reset of finished envs included"
                               + (f", + {PPO_ALLREDUCES_PER_ITER} x {GRAD_BUCKET_FLOATS * 4 / 1e6:.2f} MB gradient all-reduces per {PPO_STEPS_PER_ITER} steps ({backend})" if a.ppo else ""),
                   "envs_per_gpu": n_local, "total_envs": n_total, "dynamics": dynamics_on, "parallelism": f"env-shard x{world}",
                   "launch": "hipGraph (step + reset_done)" if a.graph else "stream launches",
                   # what the library handle resolved to: dynamics kernel, envs per block, collision set, contact parameters, manifold
                   # period and margins, curriculum path, and every developer switch it was created with (none in the product)
                   "library": env.describe(), "library_abi": int(env._lib.parc_abi_version()), "library_build_flags": build[1], "library_csrc_sha16": build[0]},
        "roofline": roof,
    }
    timeouts = int(env._lib.parc_env_dynamics_timeouts(env._handle)) if dynamics_on else 0
    out["dynamics_timeouts"] = timeouts   # LDS-flag hand-offs of k_dynamics_wave that hit their bound: must be 0
    out["dynamics_manifold_drops"] = env.dynamics_manifold_drops() if dynamics_on else 0   # contact planes that found no slot: expected 0
    if a.ppo:
        ms = [e0.elapsed_time(e1) for e0, e1 in coll_events]
        out["collective"] = {"backend": backend, "world": world, "allreduce_bytes": GRAD_BUCKET_FLOATS * 4, "allreduces_per_iter": PPO_ALLREDUCES_PER_ITER,
                             "iters_timed": len(ms), "ms_per_iter_leg": (sum(ms) / len(ms)) if ms else None,
                             "note": "40 all-reduces + the 1/world scale of the flat fp32 gradient bucket, torch.cuda events on the launch stream"
                                     + ("; one rank: RCCL's single-rank path (a device copy), exercised so the library is loaded and timed" if world == 1 else "")}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(env, a.cpu_seconds, dynamics_on, actions)
    if rank == 0:
        sys.stdout = sys.__stdout__
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if timeouts:
        sys.stderr.write(f"bench.py: rank {rank}: {timeouts} dynamics hand-off timeout(s): the measurement is void\n")
        sys.exit(5)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:  # not under torchrun: be our own launcher (no GPU call in this process)
        ngpu = _count_gpus()
        extra = {}
        if ngpu < a.gpus:
            if not os.environ.get("PARC_BENCH_SHARE_GPU"):
                sys.exit(f"bench.py: --gpus {a.gpus} but {ngpu} GPU(s) visible (set PARC_BENCH_SHARE_GPU=1 to rehearse all ranks on one GPU)")
        rc, out0 = launch_ranks(a.gpus, sys.argv[1:], extra_env=extra)
        js = [ln for ln in out0.splitlines() if ln.startswith("{")]  # (gloo's C++ side prints a connection notice to stdout)
        sys.stdout.write((js[-1] + "\n") if js else out0)
        sys.exit(rc)
    worker(a)


if __name__ == "__main__":
    main()
