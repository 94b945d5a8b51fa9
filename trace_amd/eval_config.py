"""Evaluation-side configuration of the scene-editing driver: the reference's eval JSONs
(``configs/eval/{orca,nusc}/*.json``) load unchanged into ``SceneEditingConfig``.

Mirrors (behaviour, names and defaults; written independently):
  * ``Dict``                  -- /root/reference/tbsim/configs/config.py:14-190 (attribute access on nested
                                 dicts, auto-vivifying reads that attach on first write, recursive
                                 ``update``, ``lock``/``unlock``, ``dump``, ``to_dict``)
  * ``EvaluationConfig``      -- /root/reference/tbsim/configs/eval_config.py:7-87
  * ``SceneEditingConfig``    -- /root/reference/tbsim/configs/scene_edit_config.py:11-138
  * ``finalize_eval_config``  -- the CLI post-processing of scripts/scene_editor.py:364-412
"""
import copy
import json
import os


class Dict(dict):
    """Nested dict with attribute access.  Reading a missing key of an unlocked node returns a
    detached child that attaches itself to its parent the first time something is stored in it;
    a locked node raises ``KeyError`` for unknown keys (reads and writes)."""

    def __init__(self, *args, **kwargs):
        object.__setattr__(self, "_cfg_locked", False)
        object.__setattr__(self, "_cfg_origin", kwargs.pop("__origin", None))
        for a in args:
            if a:
                for k, v in (a.items() if isinstance(a, dict) else a):
                    self[k] = self._wrap(v)
        for k, v in kwargs.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Dict):
            return Dict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(e) for e in v)
        return v

    def __getattr__(self, name):
        if name.startswith("_cfg_"):
            raise AttributeError(name)
        return self[name]

    def __setattr__(self, name, value):
        if hasattr(type(self), name):
            raise AttributeError("'%s' is a method of the config class and cannot be a key" % name)
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __missing__(self, name):
        if object.__getattribute__(self, "_cfg_locked"):
            raise KeyError(name)
        return Dict(__origin=(self, name))

    def __setitem__(self, name, value):
        if object.__getattribute__(self, "_cfg_locked") and name not in self.keys():
            raise KeyError(name)
        dict.__setitem__(self, name, self._wrap(value))
        origin = object.__getattribute__(self, "_cfg_origin")
        if origin is not None:
            object.__setattr__(self, "_cfg_origin", None)
            origin[0][origin[1]] = self

    def update(self, *args, **kwargs):
        other = {}
        if args:
            if len(args) > 1:
                raise TypeError("update takes at most one positional argument")
            other.update(args[0])
        other.update(kwargs)
        for k, v in other.items():
            if k in self and isinstance(self[k], dict) and isinstance(v, dict):
                self[k].update(v)
            else:
                self[k] = v

    def lock(self, should_lock=True):
        object.__setattr__(self, "_cfg_locked", should_lock)
        for v in self.values():
            if isinstance(v, Dict):
                v.lock(should_lock)

    def unlock(self):
        self.lock(False)

    def to_dict(self):
        def plain(v):
            if isinstance(v, Dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return plain(self)

    def dump(self, filename=None):
        text = json.dumps(self.to_dict(), indent=4)
        if filename is not None:
            with open(filename, "w") as f:
                f.write(text)
        return text

    def __repr__(self):
        return self.dump()

    def __deepcopy__(self, memo):
        other = type(self).__new__(type(self))
        dict.__init__(other)
        object.__setattr__(other, "_cfg_locked", False)
        object.__setattr__(other, "_cfg_origin", None)
        memo[id(self)] = other
        for k, v in self.items():
            dict.__setitem__(other, copy.deepcopy(k, memo), copy.deepcopy(v, memo))
        object.__setattr__(other, "_cfg_locked", object.__getattribute__(self, "_cfg_locked"))
        return other

    def clone(self):
        return copy.deepcopy(self)


class EvaluationConfig(Dict):
    def __init__(self):
        super().__init__()
        self.name = None
        self.env = "trajdata"
        self.dataset_path = None
        self.eval_class = ""
        self.seed = 0
        self.num_scenes_per_batch = 1
        self.num_scenes_to_evaluate = 1
        self.num_episode_repeats = 1
        self.start_frame_index_each_episode = None
        self.seed_each_episode = None
        self.ego_only = False
        self.ckpt_root_dir = "checkpoints/"
        self.experience_hdf5_path = None
        self.results_dir = "results/"
        self.ckpt.policy.ckpt_dir = None
        self.ckpt.policy.ckpt_key = None
        self.policy.num_action_samples = 10
        self.metrics.compute_analytical_metrics = True
        self.trajdata.trajdata_cache_location = "~/.unified_data_cache"
        self.trajdata.trajdata_rebuild_cache = False
        self.trajdata.trajdata_source_test = ["orca_maps-test"]
        self.trajdata.trajdata_data_dirs = {"orca_maps": "./datasets/orca_sim",
                                            "orca_no_maps": "./datasets/orca_sim"}
        self.trajdata.num_scenes_to_evaluate = 200
        self.trajdata.eval_scenes = list(range(200))
        self.trajdata.n_step_action = 1
        self.trajdata.num_simulation_steps = 100
        self.trajdata.skip_first_n = 0


class SceneEditingConfig(EvaluationConfig):
    def __init__(self):
        super().__init__()
        self.name = "scene_edit_eval"
        self.eval_class = "Diffuser"
        self.env = "trajdata"
        self.results_dir = "scene_edit_eval/"
        self.num_scenes_per_batch = 1
        self.policy.num_action_samples = 10
        self.policy.guide_as_filter_only = False
        self.policy.guide_with_gt = False
        self.policy.class_free_guide_w = 0.0
        self.policy.guide_clean = True
        self.metrics.compute_analytical_metrics = True
        self.trajdata.num_sim_per_scene = 1
        self.trajdata.num_scenes_to_evaluate = 100
        self.trajdata.eval_scenes = list(range(100))
        self.trajdata.n_step_action = 50
        self.trajdata.num_simulation_steps = 50
        self.trajdata.skip_first_n = 0
        self.edits.editing_source = ["heuristic"]
        self.edits.guidance_config = []
        self.edits.heuristic_config = [
            {"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
            {"name": "map_collision", "weight": 10.0, "params": {"num_points_lw": (10, 10)}},
            {"name": "target_pos_at_time", "weight": 30000.0, "params": {"target_time": 40}},
        ]


def finalize_eval_config(cfg, config_file=None, eval_class=None, policy_ckpt_dir=None, policy_ckpt_key=None,
                         results_root_dir=None, num_scenes_per_batch=None, seed=None, prefix=None,
                         editing_source=None):
    """What ``scripts/scene_editor.py:364-412`` does between parsing flags and running: JSON overlay, flag
    overrides, results dir, env sub-dict hoisted to the top level, lock."""
    if config_file is not None:
        with open(config_file) as f:
            cfg.update(**json.load(f))
    if eval_class is not None:
        cfg.eval_class = eval_class
    if policy_ckpt_dir is not None:
        assert policy_ckpt_key is not None, "Please specify a key to look for the checkpoint, e.g., 'iter50000'"
        cfg.ckpt.policy.ckpt_dir = policy_ckpt_dir
        cfg.ckpt.policy.ckpt_key = policy_ckpt_key
    if num_scenes_per_batch is not None:
        cfg.num_scenes_per_batch = num_scenes_per_batch
    if cfg.name is None:
        cfg.name = cfg.eval_class
    if prefix is not None:
        cfg.name = prefix + cfg.name
    if seed is not None:
        cfg.seed = seed
    cfg.results_dir = os.path.join(results_root_dir if results_root_dir is not None else cfg.results_dir, cfg.name)
    assert cfg.env == "trajdata", "trajdata is the only supported environment"
    if editing_source is not None:
        cfg.edits.editing_source = editing_source
    if not isinstance(cfg.edits.editing_source, list):
        cfg.edits.editing_source = [cfg.edits.editing_source]
    cfg.experience_hdf5_path = os.path.join(cfg.results_dir, "data.hdf5")
    for k in list(cfg[cfg.env].keys()):
        cfg[k] = cfg[cfg.env][k]
    cfg.pop("trajdata")
    cfg.lock()
    return cfg
