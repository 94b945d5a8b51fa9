#!/usr/bin/env python
"""Scene-editing evaluation driver with the reference's command line
(/root/reference/scripts/scene_editor.py:257-421: same flags, same config post-processing, same
loop over scene batches -> heuristic guidance -> guided rollout -> stats.json), running the
MI355X sampler behind ``--eval_class Diffuser``.

    python scripts/scene_editor.py --config_file <configs/eval/orca/target_pos.json> \
        --results_root_dir out --num_scenes_per_batch 8 --allow_random_init

Datasets and checkpoints are not available offline, so the environment is the synthetic
device-resident ``SceneBatchEnv`` (``--synthetic_*`` flags shape it; the reference's
``EnvUnifiedBuilder`` would go here) and ``--allow_random_init`` substitutes seeded random weights when
no checkpoint is found under ``ckpt_root_dir``.  Rendering and hdf5 dumps are not built.
"""
import argparse
import importlib
import json
import os
import random
import sys
from pprint import pprint

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from trace_amd.eval_config import Dict, SceneEditingConfig, finalize_eval_config      # noqa: E402
from trace_amd.policy import RolloutWrapper                                           # noqa: E402
from trace_amd.scene_edit import (compute_heuristic_guidance, guided_rollout,         # noqa: E402
                                  merge_guidance_configs)
from trace_amd.sim_env import CollisionRate, OffRoadRate, SceneBatchEnv               # noqa: E402


def run_scene_editor(eval_cfg, save_cfg, env_args, composer_args, use_gt=False, policy_module="trace_amd.policy"):
    np.random.seed(eval_cfg.seed)
    random.seed(eval_cfg.seed)
    torch.manual_seed(eval_cfg.seed)
    print("saving results to {}".format(eval_cfg.results_dir))
    os.makedirs(eval_cfg.results_dir, exist_ok=True)
    if save_cfg:
        with open(os.path.join(eval_cfg.results_dir, "config.json"), "w") as f:
            json.dump(eval_cfg.to_dict(), f)
    device = torch.device("cuda:0")

    # the policy is selected by name, as in the reference (scene_editor.py:48-52)
    composer_class = getattr(importlib.import_module(policy_module), eval_cfg.eval_class)
    policy, exp_config = composer_class(eval_cfg, device, **composer_args).get_policy()
    exp_config = Dict(exp_config)
    policy_model = policy.model
    algo_config = policy_model.algo_config
    if use_gt:
        # the recorded future instead of the model (scene_editor.py:53-58)
        from trace_amd.policy import GTNaNPolicy
        policy, policy_model = GTNaNPolicy(device=device), None
        print("WARNING: Using GT data as the policy instead of the provided model!!")
    rollout_policy = RolloutWrapper(agents_policy=policy)

    metrics = {"all_collision_rate": CollisionRate(), "all_off_road_rate": OffRoadRate()} \
        if eval_cfg.metrics.compute_analytical_metrics else {}
    env = SceneBatchEnv(algo_config, device=device, seed=eval_cfg.seed,
                        num_scenes=eval_cfg.num_scenes_per_batch,
                        num_simulation_steps=eval_cfg.num_simulation_steps, metrics=metrics, **env_args)

    heuristic_config = None
    if "heuristic" in eval_cfg.edits.editing_source:
        heuristic_config = eval_cfg.edits.heuristic_config if eval_cfg.edits.heuristic_config is not None else []
        heuristic_config = [h.to_dict() if isinstance(h, Dict) else h for h in heuristic_config]

    result_stats = None
    scene_i = 0
    n_eval = min(eval_cfg.num_scenes_to_evaluate, env.total_num_scenes)
    eval_scenes = [s for s in eval_cfg.eval_scenes if s < env.total_num_scenes]
    while scene_i < n_eval:
        scene_indices = eval_scenes[scene_i: scene_i + eval_cfg.num_scenes_per_batch]
        scene_i += eval_cfg.num_scenes_per_batch
        scenes_valid = env.reset(scene_indices=scene_indices, start_frame_index=None)
        scene_indices = [si for si, ok in zip(scene_indices, scenes_valid) if ok]
        if not scene_indices:
            continue
        start_frames = [exp_config.algo.history_num_frames + 1] * len(scene_indices)
        guidance_config = None
        if "config" in eval_cfg.edits.editing_source:
            guidance_config = eval_cfg.edits.guidance_config
        if "heuristic" in eval_cfg.edits.editing_source:
            heur = compute_heuristic_guidance(heuristic_config, env, scene_indices, start_frames)
            if len(heuristic_config) > 0:
                valid = [i for i, c in enumerate(heur) if len(c) > 0]
                heur = [heur[i] for i in valid]
                scene_indices = [scene_indices[i] for i in valid]
                start_frames = [start_frames[i] for i in valid]
                if not scene_indices:
                    print("No scenes with valid heuristic configs in this batch, skipping...")
                    continue
            guidance_config = merge_guidance_configs(guidance_config, heur)
        stats, info = guided_rollout(env, rollout_policy, policy_model, n_step_action=eval_cfg.n_step_action,
                                     guidance_config=guidance_config, scene_indices=scene_indices,
                                     obs_to_torch=True, horizon=eval_cfg.num_simulation_steps, use_gt=use_gt,
                                     start_frames=start_frames)
        print(info["scene_index"])
        pprint(stats)
        if result_stats is None:
            result_stats = dict(stats)
            result_stats["scene_index"] = np.array(info["scene_index"])
        else:
            for k in stats:
                result_stats[k] = np.concatenate([result_stats[k], stats[k]], 0) if k in result_stats else stats[k]
            result_stats["scene_index"] = np.concatenate([result_stats["scene_index"], np.array(info["scene_index"])])
        with open(os.path.join(eval_cfg.results_dir, "stats.json"), "w") as fp:
            json.dump({k: np.asarray(v).tolist() for k, v in result_stats.items()}, fp)
    return result_stats


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", type=str, default=None, help="A json file containing evaluation configs")
    p.add_argument("--eval_class", type=str, default=None)
    p.add_argument("--policy_ckpt_dir", type=str, default=None)
    p.add_argument("--policy_ckpt_key", type=str, default=None)
    p.add_argument("--results_root_dir", type=str, default=None)
    p.add_argument("--num_scenes_per_batch", type=int, default=None)
    p.add_argument("--render", action="store_true", default=False)
    p.add_argument("--render_img", action="store_true", default=False)
    p.add_argument("--render_size", type=int, default=400)
    p.add_argument("--render_px_per_m", type=float, default=12.0)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--prefix", type=str, default=None)
    p.add_argument("--ground_truth", action="store_true", default=False)
    p.add_argument("--editing_source", type=str, choices=["config", "heuristic", "none"], nargs="+")
    # additions of this build (no datasets / checkpoints offline)
    p.add_argument("--allow_random_init", action="store_true")
    p.add_argument("--synthetic_scenes", type=int, default=16, help="scenes in the synthetic pool")
    p.add_argument("--synthetic_agents", type=int, default=32, help="agents per scene")
    p.add_argument("--synthetic_dense", action="store_true")
    p.add_argument("--num_scenes_to_evaluate", type=int, default=None)
    args = p.parse_args(argv)
    assert not (args.render or args.render_img), "rendering is not built (SURVEY.md section 2: out of scope)"

    cfg = SceneEditingConfig()
    if args.config_file is not None:
        with open(args.config_file) as f:
            cfg.update(**json.load(f))
    if args.num_scenes_to_evaluate is not None:
        cfg.trajdata.num_scenes_to_evaluate = args.num_scenes_to_evaluate
    cfg = finalize_eval_config(cfg, eval_class=args.eval_class, policy_ckpt_dir=args.policy_ckpt_dir,
                               policy_ckpt_key=args.policy_ckpt_key, results_root_dir=args.results_root_dir,
                               num_scenes_per_batch=args.num_scenes_per_batch, seed=args.seed, prefix=args.prefix,
                               editing_source=args.editing_source)
    env_args = dict(agents_per_scene=[args.synthetic_agents] * args.synthetic_scenes, dense=args.synthetic_dense)
    composer_args = dict(allow_random_init=args.allow_random_init, seed=cfg.seed, random_map_encoder=True)
    return run_scene_editor(cfg, save_cfg=True, env_args=env_args, composer_args=composer_args,
                            use_gt=args.ground_truth)


if __name__ == "__main__":
    main()
