"""Shared helpers for the test-suite: bundled configs -> Map / CarParams / Camera, golden loaders."""
import copy
import os

import numpy as np
import yaml

from tinycarlo_amd.camera import Camera
from tinycarlo_amd.config import CarParams, bundled_config
from tinycarlo_amd.map import Map

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RES = {"r64": [64, 64], "r128": [128, 128], "r480": [480, 640]}
CFG = {"simple_layout": "config_simple_layout.yaml", "knuffingen": "config_knuffingen.yaml",
       "formula_student_track": "config_formula_student_track.yaml"}

_cache = {}


# configs that are test inputs only (tests/golden/): the synthetic maps of make_stress_map.py
TEST_CFG = {"stress_graph": "config_stress_graph.yaml", "oneway": "config_oneway.yaml"}
# random maps the reference was run on (gen_golden.py fuzz): fuzz2000 .. fuzz2007
FUZZ_MAPS = sorted(f[len("config_"):-len(".yaml")] for f in os.listdir(GOLDEN) if f.startswith("config_fuzz") and f.endswith(".yaml"))
TEST_CFG.update({n: f"config_{n}.yaml" for n in FUZZ_MAPS})


def load_cfg(map_name):
    path = os.path.join(GOLDEN, TEST_CFG[map_name]) if map_name in TEST_CFG else bundled_config(CFG[map_name])
    with open(path) as f:
        return yaml.safe_load(f), path


def setup(map_name, res_key="r64", **cam_over):
    """-> (cfg, Map, CarParams, Camera) for a bundled map with the camera resolution overridden."""
    cfg, path = load_cfg(map_name)
    key = ("map", map_name)
    if key not in _cache:
        _cache[key] = Map(cfg["map"], base_path=path)
    m = _cache[key]
    car = CarParams.from_config(1 / cfg["sim"].get("fps", 30), cfg["car"])
    cc = copy.deepcopy(cfg["camera"])
    cc["resolution"] = list(RES[res_key])
    cc.update(cam_over)
    return cfg, m, car, Camera(cc)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rollout_files():
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith("rollout_") and f.endswith(".npz"))


def map_of(fname):
    for name in list(CFG) + list(TEST_CFG):
        if name in fname:
            return name
    raise KeyError(fname)


def cam_keys(d):
    return [k[4:] for k in d.files if k.startswith("seg_") and not k.endswith("_off")]


def wrapper_cases():
    """tests/golden/wrappers.json: the reference's wrapper classes run over recorded infos (gen_golden.py wrappers)"""
    import json
    with open(os.path.join(GOLDEN, "wrappers.json")) as f:
        return json.load(f)


def build_stack(env, spec, **extra):
    """stacks this repo's wrappers as described by a wrappers.json "spec" ([[class name, kwargs]], innermost first)"""
    import tinycarlo_amd.wrapper as W
    for cls, kw in spec:
        env = getattr(W, cls)(env, **kw, **extra)
    return env


def terms_of(spec, names):
    """the same stack as tinycarlo_amd.terms.Term objects"""
    from tinycarlo_amd import terms as T
    out = []
    for cls, kw in spec:
        if cls == "LanelineSparseRewardWrapper":
            out.append(T.laneline_sparse_reward(names, kw["sparse_rewards"]))
        elif cls == "LanelineLinearRewardWrapper":
            out.append(T.laneline_linear_reward(names, kw["max_rewards"]))
        elif cls == "CTESparseRewardWrapper":
            out.append(T.cte_sparse_reward(**kw))
        elif cls == "CTELinearRewardWrapper":
            out.append(T.cte_linear_reward(**kw))
        elif cls == "LanelineCrossingTerminationWrapper":
            out.append(T.laneline_crossing_termination(names, kw["lanelines"]))
        elif cls == "CTETerminationWrapper":
            out.append(T.cte_termination(**kw))
        elif cls == "CrashTerminationWrapper":
            out.append(T.crash_termination(kw["velcoity_threshold"], kw["number_of_steps"]))
        else:
            raise KeyError(cls)
    return out
