#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the motion-tracking env step at 65 536 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is what the PPO rollout does to the env per control step (base_agent.py:348-370): ``env.step(action)``
followed by the reset of the envs that finished.  Envs shard across ranks with no data-path collective
(``scaling: weak``: every GPU owns ``--envs`` envs; offsets follow the global env index).  Rank 0 prints ONE JSON line.

The JSON carries ``roofline`` (algorithmic HBM bytes of the step kernel / its hipEvent-measured duration, against
the 8 TB/s HBM peak) and, at N=1, ``cpu_baseline`` (the CPU oracle — a scalar C port of the reference step, first
pinned against the reference's golden vectors — timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--config", default=os.path.join(REPO, "data/configs/tracker_config/dm_env_default.yaml"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--motions", type=int, default=0,
                    help="replicate the bundled clips to this many library entries (SURVEY 8(d) cfg 3: 1024); 0 = the 5 clips as they are")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: step + reset_done as one hipGraph launch (parc_env_step_reset_graph); kernel timings are then taken from a short separate run")
    ap.add_argument("--dynamics", type=int, default=1, help="1: full step (rigid-body dynamics + contact), 0: kinematic step only")
    return ap.parse_args()


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
    clips = helpers.load_clips([c.name for c in sc.clips])
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
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} {'full (dynamics + obs/reward/done)' if dynamics_on else 'kinematic'} steps x {n} envs, C/C++ oracle "
                      "(scalar, -O2, one thread per core), same scene and state"}


def main():
    a = parse()
    import ctypes as C
    import torch
    from parc_amd import lib as L
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on cuda:0, gloo for the barrier / max
    backend = os.environ.get("PARC_BENCH_BACKEND", "nccl")
    if os.environ.get("PARC_BENCH_SHARE_GPU"):
        local = 0
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    from parc_amd.envs.hip_parkour_env import HipParkourEnv
    from parc_amd.util import path_loader
    cfg = path_loader.load_config(a.config)
    if a.motions > 0:  # synthetic large library: names suffixed, same content (symlinks), weights unchanged
        import tempfile, yaml
        src = path_loader.load_config(str(path_loader.resolve_path(cfg["env"]["dm"]["motion_file"])))["motions"]
        d = tempfile.mkdtemp(prefix=f"parc_bench_lib_r{rank}_")
        ents = []
        for i in range(a.motions):
            m = src[i % len(src)]
            f = str(path_loader.resolve_path(m["file"]))
            dst = os.path.join(d, os.path.splitext(os.path.basename(f))[0] + f"_r{i:05d}.pkl")
            os.symlink(f, dst)
            ents.append({"file": dst, "weight": m.get("weight", 1.0)})
        with open(os.path.join(d, "motions.yaml"), "w") as fh:
            yaml.safe_dump({"motions": ents}, fh)
        cfg["env"]["dm"]["motion_file"] = os.path.join(d, "motions.yaml")
        cfg["env"]["dm"].pop("terrain_save_path", None)
    dyn = bool(a.dynamics)
    sys.stdout = sys.stderr if rank == 0 else open(os.devnull, "w")  # the only stdout line is the JSON below
    env = HipParkourEnv(cfg, a.envs, dev, False, env_id_base=rank * a.envs, total_envs=world * a.envs, seed=1234 + rank,
                        mirror_ref_state=False, enable_dynamics=dyn)
    dynamics_on = bool(env._scene.cfg.enable_dynamics)
    mode = cfg["env"]["dm"].get("terrain_build_mode", "square")
    ncl = len(env._scene.clips)
    lib_desc = (f"{a.motions} library entries (the bundled clips replicated)" if a.motions > 0 else
                (f"{ncl} bundled clips" if ncl > 1 else f"clip {env._scene.clips[0].name}")) + \
               {"square": " on a square blocky grid", "wide": " on a wide blocky grid", "file": " on its own terrain"}.get(mode, "")
    D = env._char_dof_pos.shape[1]
    # untrained-policy actions (SURVEY §8(d) cfg 3): action-normalizer mean + N(0, 0.05^2) * std
    lo, hi = env._action_bound_low, env._action_bound_high
    mean, std = 0.5 * (hi + lo), 0.5 * (hi - lo)
    actions = [mean + 0.05 * std * torch.randn(a.envs, D, device=dev) for _ in range(4)]
    env.reset()

    def one_step(i):
        if a.graph:
            env.step_and_reset_done(actions[i & 3] if dynamics_on else None)
        else:
            env.step(actions[i & 3])
            env.reset_done()

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
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
    dms, pms, nst = C.c_double(), C.c_double(), C.c_int32()
    L.check(env._lib.parc_env_get_kernel_timing(env._handle, C.byref(dms), C.byref(pms), C.byref(nst)))
    L.check(env._lib.parc_env_set_kernel_timing(env._handle, 0))
    if a.graph:  # events are not part of the captured graph: time the kernels over 20 ordinary steps instead
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 1))
        for i in range(20):
            env.step(actions[i & 3]); env.reset_done()
        L.check(env._lib.parc_env_get_kernel_timing(env._handle, C.byref(dms), C.byref(pms), C.byref(nst)))
        L.check(env._lib.parc_env_set_kernel_timing(env._handle, 0))
    else:
        assert nst.value == a.steps
    dyn_ms, post_ms = float(dms.value), float(pms.value)
    # HBM traffic of k_env_post from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs;
    # FETCH_SIZE doubled per the gfx950 correction).  Only valid for the configuration it was collected on.
    post_traffic = None
    pmc_path = os.path.join(REPO, "profiles", "r01_pmc_hbm_traffic.json")
    if os.path.exists(pmc_path) and a.envs == 65536:
        pmc = json.load(open(pmc_path))
        k = "void k_env_post<0>"
        post_traffic = (2.0 * pmc["FETCH_SIZE_KiB_avg_per_dispatch"][k] + pmc["WRITE_SIZE_KiB_avg_per_dispatch"][k]) * 1024.0
    post_gbs = BYTES_KINEMATIC * a.envs / (post_ms * 1e-3) / 1e9
    if dynamics_on:   # dominant kernel = k_dynamics: state 276 + action 112 in, state 276 + contact forces 180 out
        kname = env._lib.parc_env_dynamics_kernel(env._handle).decode()
        kms, bytes_per, traffic = dyn_ms, BYTES_DYN_KERNEL, None
        dpmc = os.path.join(REPO, "profiles", "r01_pmc_dynamics.json")
        if os.path.exists(dpmc) and a.envs == 65536:
            dj = json.load(open(dpmc))
            if dj.get("kernel") == kname:
                traffic = 2.0 * dj["FETCH_SIZE_KiB"] * 1024.0 + dj["WRITE_SIZE_KiB"] * 1024.0
    else:
        kname, kms, bytes_per, traffic = "k_env_prep + k_env_post<MODE_STEP>", post_ms, BYTES_KINEMATIC, post_traffic
    achieved = bytes_per * a.envs / (kms * 1e-3) / 1e9
    out = {
        "metric": "env-steps/s", "value": a.envs * world * a.steps / dt, "unit": "env-steps/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("full step (dynamics + obs/reward/done)" if dynamics_on else "kinematic step (ref slerp + FK + 441-ray hf + obs + reward + done), no physics")
                               + f", {a.envs} envs per GPU, " + lib_desc + ", reset of finished envs included",
                   "envs_per_gpu": a.envs, "total_envs": a.envs * world, "dynamics": dynamics_on, "parallelism": f"env-shard x{world}",
                   "launch": "hipGraph (step + reset_done)" if a.graph else "stream launches"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": kname, "kernel_ms": kms, "algorithmic_bytes_per_env_step": bytes_per,
                     "note": ("the dynamics kernel is VALU-issue / latency bound (rigid-body recursion, ~47k VALU instructions per wave at 1 wave per SIMD; "
                              "profiles/r01_pmc_dynamics.json), far below the HBM ceiling by construction (SURVEY 8d); obs_kernel is the HBM-bound one")
                             if dynamics_on else "",
                     "obs_kernel": {"kernel": "k_env_prep + k_env_post<MODE_STEP>", "kernel_ms": post_ms, "achieved": post_gbs,
                                    "frac": post_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": BYTES_KINEMATIC,
                                    "traffic": post_traffic},
                     "whole_step_algorithmic_bytes": BYTES_DYNAMICS if dynamics_on else BYTES_KINEMATIC},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(env, a.cpu_seconds, dynamics_on, actions)
    if rank == 0:
        sys.stdout = sys.__stdout__
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
