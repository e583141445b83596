"""Map loader: JSON polyline graphs -> flat arrays for the HIP library.

Mirrors the reference's ``tinycarlo/map.py:9-69`` (load, px->m scaling, layer order = JSON key
order, spawn sampling) and keeps the small read-only ``Layer`` view callers of the reference
touch (``map.lanelines[i].nodes/edges/name/color``, ``map.lanepath``).  Geometry queries on the
hot path (``layer.py:33-187``) are NOT implemented here: they run inside the HIP kernels.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np


class Layer:
    """Read-only view of one polyline graph (reference ``layer.py:15-19``)."""

    def __init__(self, name: str, color: Sequence[int], nodes: List[List[float]], edges: List[List[int]]):
        self.name = name
        self.color = list(color)
        self.nodes = nodes
        self.edges = edges

    def get_edge_coordinates_list(self):
        return [(self.nodes[e[0]], self.nodes[e[1]]) for e in self.edges]

    def get_edge_coordinates(self, edge):
        return self.nodes[edge[0]], self.nodes[edge[1]]

    def get_next_nodes(self, node_idx: int) -> List[int]:
        return [e[1] for e in self.edges if e[0] == node_idx]

    def get_prev_nodes(self, node_idx: int) -> List[int]:
        return [e[0] for e in self.edges if e[1] == node_idx]


class Map:
    """``Map(map_config, base_path)`` as in the reference (``map.py:9-26``).

    ``map_config`` needs ``json_path`` and ``pixel_per_meter``; ``spawn_points`` is optional.
    ``base_path`` is the path of the YAML file the config came from (its directory is the base
    for a relative ``json_path``), or ``None`` for the current directory.
    """

    def __init__(self, map_config: Dict[str, Any], base_path: Optional[str] = None):
        self.spawn_points: Optional[List[int]] = map_config.get("spawn_points", None)
        base = "./" if base_path is None else os.path.dirname(base_path)
        map_path = os.path.join(base, map_config["json_path"])
        ppm = map_config["pixel_per_meter"]
        with open(map_path) as f:
            data = json.load(f)
        # map.py:28-37 -- every coordinate is divided by pixel_per_meter (python float division)
        data["height"] = data["height"] / ppm
        data["width"] = data["width"] / ppm
        for layer in data["lanelines"].values():
            for n in layer["nodes"]:
                n[0] = n[0] / ppm
                n[1] = n[1] / ppm
        for n in data["lanepath"]["nodes"]:
            n[0] = n[0] / ppm
            n[1] = n[1] / ppm
        self.lanelines: List[Layer] = [Layer(name, l["layer_color"], l["nodes"], l["edges"])
                                       for name, l in data["lanelines"].items()]
        lp = data["lanepath"]
        self.lanepath: Layer = Layer("lanepath", lp["layer_color"], lp["nodes"], lp["edges"])
        self.dimension: Tuple[float, float] = (data["height"], data["width"])
        self.pixel_per_meter = ppm
        self.path = map_path

    # ---- reference accessors (map.py:39-49)
    def get_laneline_names(self) -> List[str]:
        return [l.name for l in self.lanelines]

    def get_lanelines(self):
        return [l.get_edge_coordinates_list() for l in self.lanelines]

    def get_laneline_nodes(self):
        return [l.nodes for l in self.lanelines]

    def get_laneline_edges(self):
        return [l.edges for l in self.lanelines]

    def get_lanepath(self):
        return self.lanepath.get_edge_coordinates_list()

    def get_laneline_colors(self):
        return [l.color for l in self.lanelines]

    # ---- spawn sampling (map.py:51-69), index only: the pose itself is set by the reset kernel
    def sample_spawn_node(self, np_random: np.random.Generator) -> int:
        """Draws a spawn node exactly like ``Map.sample_spawn`` consumes ``np_random``:
        ``choice(spawn_points)`` or ``integers(0, len(nodes)-1)`` (the last node is never drawn),
        re-drawing while the node has no outgoing lanepath edge."""
        has_next = self._has_next()
        for _ in range(100000):
            if self.spawn_points is None:
                idx = int(np_random.integers(0, len(self.lanepath.nodes) - 1, size=1, dtype=int)[0])
            else:
                idx = int(np_random.choice(self.spawn_points))
            if has_next[idx]:
                return idx
        raise RuntimeError("no spawnable lanepath node found")

    def spawn_table(self) -> np.ndarray:
        """The candidates of ``Map.sample_spawn`` (map.py:61: ``spawn_points``, or node ids ``0..len-2``) that have an
        out-edge, duplicates kept: a uniform draw from it is distributed like the draw-again-on-sinks loop of
        map.py:62-64.  Used by the device-side spawn sampling (``tc_env_set_spawn_table``)."""
        has_next = self._has_next()
        cand = range(len(self.lanepath.nodes) - 1) if self.spawn_points is None else [int(p) for p in self.spawn_points]
        tab = np.array([c for c in cand if has_next[c]], dtype=np.int32)
        if tab.size == 0:
            raise RuntimeError("no spawnable lanepath node found")
        return tab

    def _has_next(self) -> np.ndarray:
        hn = getattr(self, "_hn", None)
        if hn is None:
            hn = np.zeros(len(self.lanepath.nodes), dtype=bool)
            for e in self.lanepath.edges:
                hn[e[0]] = True
            self._hn = hn
        return hn

    # ---- flat arrays for tc_map_create (include/tinycarlo_hip.h)
    def flat(self) -> Dict[str, np.ndarray]:
        node_count = np.array([len(l.nodes) for l in self.lanelines], dtype=np.int32)
        edge_count = np.array([len(l.edges) for l in self.lanelines], dtype=np.int32)
        nodes = np.array([n for l in self.lanelines for n in l.nodes], dtype=np.float64).reshape(-1, 2)
        edges = np.array([e for l in self.lanelines for e in l.edges], dtype=np.int32).reshape(-1, 2)
        colors = np.array([l.color for l in self.lanelines], dtype=np.uint8).reshape(-1, 3)
        lp_nodes = np.array(self.lanepath.nodes, dtype=np.float64).reshape(-1, 2)
        lp_edges = np.array(self.lanepath.edges, dtype=np.int32).reshape(-1, 2)
        return dict(node_count=node_count, edge_count=edge_count, nodes=np.ascontiguousarray(nodes),
                    edges=np.ascontiguousarray(edges), colors=np.ascontiguousarray(colors),
                    lp_nodes=np.ascontiguousarray(lp_nodes), lp_edges=np.ascontiguousarray(lp_edges))
