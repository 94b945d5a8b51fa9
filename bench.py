#!/usr/bin/env python
"""Benchmark of the TRACE guided sampler on MI355X (contract: see the round prompt / DESIGN.md).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full sampling call of the hot path over one batch: 100 DDPM steps x 52-step
horizon through ``DiffuserTrafficModel.get_action`` (conditioning, denoiser forward + input-VJP,
guidance, posterior/dynamics, sample selection), inputs resident in HBM.  Workload = BASELINE.json
configs[1]: target_pos guidance + classifier-free guidance (w=0.5), 256 agents x 20 samples,
clean guidance, per GPU (weak scaling: every rank owns its own scenes; the only exchange is the
all-gather of scores / selected plans at the end of each call).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trace_amd import synthetic as syn                    # noqa: E402
from trace_amd.config import ModelConfig                  # noqa: E402
from trace_amd.weights import random_state_dict           # noqa: E402

WORKLOADS = {
    # name: (scenes, agents per scene, samples, guidance kind, cfg w, clean)
    "cfg2_target_pos_cfg_clean": dict(scenes=8, agents=32, samples=20, guide="target_pos", w=0.5,
                                      clean=True, dense=False),
    "cfg1_unguided": dict(scenes=1, agents=16, samples=20, guide=None, w=0.0, clean=False, dense=False),
    "cfg3_dense_collision": dict(scenes=1, agents=128, samples=20, guide="collision", w=0.0,
                                 clean=True, dense=True),
    # BASELINE configs[3] per GPU at 8 GPUs: 64 scenes x 32 agents x 20 samples, guidance evaluated once on
    # the final samples and used for selection only (guide_as_filter_only), scores all-gathered
    "cfg4_filter_only": dict(scenes=64, agents=32, samples=20, guide="target_pos+collision", w=0.0,
                             clean=True, dense=False, filter_only=True),
    # BASELINE configs[4] per GPU at 8 GPUs: 128 scenes x 32 agents, closed loop on the device-resident
    # environment, replanning every 10 frames for 100 frames, global_target_pos + collision guidance, 10 samples
    "cfg5_closed_loop": dict(scenes=128, agents=32, samples=10, guide="closed_loop", w=0.0, clean=True,
                             dense=False, closed_loop=True, n_step_action=10, sim_steps=100),
}

CLOSED_LOOP_HEURISTICS = [
    {"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
    {"name": "map_collision", "weight": 10.0, "params": {"num_points_lw": [10, 10]}},
    {"name": "global_target_pos", "weight": 10000.0,
     "params": {"target_time": 100, "urgency": 0.7, "pref_speed": 1.25, "min_progress_dist": 0.5, "dt": 0.1}},
]


def build_inputs(cfg, wl, seed, device):
    aps = [wl["agents"]] * wl["scenes"]
    batch = syn.make_batch(cfg, aps, seed=seed, dense=wl["dense"], max_neighbors=16)
    guidance = None
    if wl["guide"] == "target_pos":
        guidance = syn.target_pos_guidance(batch, aps, weight=30000.0, seed=seed)
    elif wl["guide"] == "collision":
        guidance = syn.collision_guidance(aps, agent_weight=1000.0, map_weight=10.0, num_disks=1)
    elif wl["guide"] == "target_pos+collision":
        guidance = syn.merge_guidance(syn.target_pos_guidance(batch, aps, weight=30000.0, seed=seed),
                                      syn.collision_guidance(aps, agent_weight=1000.0, map_weight=10.0, num_disks=1))
    if device is not None:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return aps, batch, guidance


def cpu_baseline(cfg, sd, wl_name, seconds_hint=20.0):
    """The CPU oracle (port of the reference's algorithm, oracle/trace_oracle.py) timed on this
    host's cores on a bounded sample of the same workload: 16 agents x 20 samples, a few of the
    100 DDPM steps, scaled to a full call."""
    from oracle import trace_oracle as orc
    wl = dict(WORKLOADS[wl_name])
    wl["scenes"], wl["agents"] = 1, 16
    aps, batch, guidance = build_inputs(cfg, wl, 1234, None)
    o = orc.OracleSampler(cfg, sd)
    o.set_guidance(guidance)
    N = wl["samples"]
    M = sum(aps) * N
    # warm-up (thread pools, oneDNN primitive caches), then size the sample for ~seconds_hint
    o.sample(batch, N, wl["w"], True, wl["clean"], n_steps=1, return_diffusion=False)
    t0 = time.perf_counter()
    o.sample(batch, N, wl["w"], True, wl["clean"], n_steps=2, return_diffusion=False)
    per_step = (time.perf_counter() - t0) / 2
    n_steps = int(max(4, min(cfg.n_diffusion_steps, seconds_hint / max(per_step, 1e-3))))
    t0 = time.perf_counter()
    o.sample(batch, N, wl["w"], True, wl["clean"], n_steps=n_steps, return_diffusion=False)
    dt = time.perf_counter() - t0
    full = dt * cfg.n_diffusion_steps / n_steps
    return {"value": M / full, "unit": "trajectories/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d agents x %d samples (M=%d), %d of %d DDPM steps timed (%.1f s), scaled to a "
                      "full call" % (sum(aps), N, M, n_steps, cfg.n_diffusion_steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_target_pos_cfg_clean", choices=sorted(WORKLOADS))
    ap.add_argument("--scenes", type=int, default=None, help="override scenes per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step loop as one hipGraph (captured during warm-up); pays off on "
                         "launch-bound small batches such as cfg1")
    ap.add_argument("--host-inputs", action="store_true",
                    help="keep the observation batch in host memory (PCIe-inclusive rate; not the headline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # TRACE_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
    # (ranks then share devices; RCCL itself refuses two ranks on one GPU)
    backend = os.environ.get("TRACE_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    from trace_amd import _cabi
    from trace_amd.distributed import gather_actions
    from trace_amd.policy import DiffuserTrafficModel, PolicyWrapper

    cfg = ModelConfig()
    sd = random_state_dict(cfg, 0)
    wl = dict(WORKLOADS[args.workload])
    if args.scenes is not None:
        wl["scenes"] = args.scenes
    N = wl["samples"]
    gen = torch.Generator(device=device).manual_seed(7 + rank)
    calls_per_step = 1
    if wl.get("closed_loop"):
        # the replanning loop: device-resident environment -> observation -> sampler -> step, scenes of
        # this rank only (scene ids offset by rank so every rank simulates different scenes)
        from trace_amd.map_encoder import random_map_encoder_state_dict
        from trace_amd.policy import RolloutWrapper
        from trace_amd.scene_edit import compute_heuristic_guidance, guided_rollout
        from trace_amd.sim_env import SceneBatchEnv
        aps = [wl["agents"]] * wl["scenes"]
        B = sum(aps)
        M = B * N
        sd = dict(sd)
        sd.update(random_map_encoder_state_dict(cfg.image_shape[0], cfg.map_grid_feature_dim, cfg.image_shape[1:], seed=0))
        model = DiffuserTrafficModel(cfg, sd, device)
        policy = PolicyWrapper.wrap_controller(model, num_action_samples=N, class_free_guide_w=wl["w"],
                                               guide_as_filter_only=False, guide_with_gt=False,
                                               guide_clean=wl["clean"])
        env = SceneBatchEnv(cfg, aps, device=device, seed=100 + rank, num_simulation_steps=wl["sim_steps"])
        rollout = RolloutWrapper(agents_policy=policy)
        _inner_get_action = rollout.get_action

        def _get_action_with_progress(obs, step_index, **kw):      # a long run must keep printing
            print("[bench] replanning at frame %d" % step_index, file=sys.stderr, flush=True)
            return _inner_get_action(obs, step_index, **kw)
        rollout.get_action = _get_action_with_progress
        # the first map-encoder call compiles MIOpen kernels for minutes without output: keep a heartbeat
        import threading

        def _heartbeat():
            while True:
                time.sleep(45)
                print("[bench] alive", file=sys.stderr, flush=True)
        threading.Thread(target=_heartbeat, daemon=True).start()
        scene_ids = list(range(len(aps)))
        guidance = compute_heuristic_guidance(CLOSED_LOOP_HEURISTICS, env, scene_ids, None)
        calls_per_step = wl["sim_steps"] // wl["n_step_action"]

        def one_step(i):
            stats, _ = guided_rollout(env, rollout, model, n_step_action=wl["n_step_action"],
                                      guidance_config=guidance, scene_indices=scene_ids,
                                      horizon=wl["sim_steps"], policy_kwargs={"generator": gen})
            obs = env.get_observation()["agents"]
            return (obs["centroid"], stats)
    else:
        aps, batch, guidance = build_inputs(cfg, wl, 100 + rank, None if args.host_inputs else device)
        B = sum(aps)
        M = B * N
        model = DiffuserTrafficModel(cfg, sd, device)
        policy = PolicyWrapper.wrap_controller(model, num_action_samples=N, class_free_guide_w=wl["w"],
                                               guide_as_filter_only=bool(wl.get("filter_only")), guide_with_gt=False,
                                               guide_clean=wl["clean"])
        if guidance is not None:
            model.set_guidance(guidance, batch)

        def one_step(i):
            action, info = policy.get_action(batch, step_index=i, generator=gen, use_graph=args.graph)
            scores = info.get("guide_scores")
            if scores is None:
                scores = torch.zeros(B, N, device=device)
            # the path's one exchange step (RCCL all-gather; a no-op on one GPU)
            return gather_actions(action.positions, action.yaws, scores, info["act_idx"])

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out[0]).all()), "non-finite plan"

    roofline = None
    if not args.no_roofline and rank == 0 and not wl.get("closed_loop"):
        # dominant kernel = conv_gemm_kernel (denoiser forward + input-VJP): per-launch HIP events on
        # the launch stream over a short profiled stretch of the same workload
        L = _cabi.lib()
        n_prof_steps = 5
        _cabi.check(L.trace_profile_begin(200 * n_prof_steps), "trace_profile_begin")
        model.sampler(batch, N, class_free_guide_w=wl["w"], guide_clean=wl["clean"], n_steps=n_prof_steps,
                      generator=gen)
        ms, nl, fl = C.c_double(), C.c_int64(), C.c_double()
        _cabi.check(L.trace_profile_end(C.byref(ms), C.byref(nl), C.byref(fl)), "trace_profile_end")
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # The conv kernel computes fp32 results on the bf16 matrix pipe: every algorithmic multiply-add
        # is 6 bf16 MFMA multiply-adds (exact 3-way operand split), so its MFMA
        # roof in ALGORITHMIC FLOP/s is the dense bf16 peak / 6.  `achieved` counts algorithmic FLOPs.
        bf16_peak, fp32_mfma_peak = 2500.0, 157.3
        peak = bf16_peak / 6.0
        # HBM bytes per launch cannot be counted from inside this process: the value comes from the
        # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/, corrected as
        # MI355X_MICROARCH.md prescribes), null when that summary is absent.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = float(json.load(f)["hbm_bytes_per_launch"])
        roofline = {"bound": "mfma",
                    "kernel": "conv_bf3_kernel (fp32 implicit-GEMM conv as 6 bf16-MFMA terms per product, "
                              "fp32 accumulate; a few stride-2 layers still on the fp32-MFMA conv_gemm_kernel)",
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "peak_is": "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMA terms per algorithmic product",
                    "frac_of_fp32_mfma_peak": achieved / fp32_mfma_peak,
                    "traffic": traffic, "launches": int(nl.value),
                    "avg_launch_us": ms.value * 1e3 / max(nl.value, 1),
                    "conv_ms_per_ddpm_step": ms.value / n_prof_steps}
    if world > 1:
        dist.barrier()

    if rank == 0:
        total = M * world * args.steps * calls_per_step
        res = {
            "metric": "guided trajectory samples/sec (100 DDPM steps, 52-step horizon)",
            "value": total / elapsed, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 results everywhere; the denoiser GEMMs evaluate each fp32 product as an exact 3-way "
                          "bf16 operand split (6 bf16-MFMA terms, fp32 accumulate; rms error equal to the fp32 "
                          "MFMA's, DESIGN.md section 5.1)",
            "data": "synthetic (seeded ORCA-style scenes, random-init orca_mixed weights)"
                    + ("; inputs in HOST memory, PCIe copy inside the timed region" if args.host_inputs else ""),
            "config": {"workload": args.workload, "agents_per_gpu": B, "samples_per_agent": N,
                       "trajectories_per_gpu": M, "ddpm_steps": cfg.n_diffusion_steps,
                       "horizon": cfg.horizon, "class_free_guide_w": wl["w"],
                       "guidance": wl["guide"], "guide_clean": wl["clean"], "hip_graph": bool(args.graph),
                       "guide_as_filter_only": bool(wl.get("filter_only")),
                       "sampling_calls_per_step": calls_per_step,
                       "parallelism": "scenes sharded over %d GPU(s), all-gather of scores/plans" % world},
        }
        if roofline is not None:
            res["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline and not wl.get("closed_loop"):
            res["cpu_baseline"] = cpu_baseline(cfg, sd, args.workload)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
