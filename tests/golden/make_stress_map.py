#!/usr/bin/env python3
"""Writes tests/golden/stress_graph.json: a synthetic map in the reference's JSON schema whose graphs contain what
the bundled maps do not -- a lanepath hub with five successors and five predecessors, a dead end, a self-loop, a
duplicate edge, and lane-line layers with zero-length, vertical, horizontal and duplicate edges, a self-loop layer and
an isolated node.  It is INPUT data for the reference (gen_golden.py runs the reference on it) and for the tests;
deterministic, no RNG."""
import json
import math
import os

OUT = os.path.dirname(os.path.abspath(__file__))
W, H = 900, 600
cx, cy = 450, 300


def ring(rx, ry, n, phase=0.0, reverse=False):
    pts = []
    for i in range(n):
        a = 2 * math.pi * (i / n) + phase
        if reverse:
            a = -a
        pts.append([int(round(cx + rx * math.cos(a))), int(round(cy + ry * math.sin(a)))])
    return pts


def closed(n, off=0):
    return [[off + i, off + (i + 1) % n] for i in range(n)]


# ---- lanepath: outer lane (counter-clockwise in image coords), inner lane the other way round, a hub in the middle
n_o, n_i = 36, 30
nodes = ring(330, 200, n_o) + ring(290, 165, n_i, reverse=True)
edges = closed(n_o) + closed(n_i, n_o)
hub = len(nodes)
nodes.append([cx, cy])
into = [n_o + k for k in (0, 6, 12, 18, 24)]       # five inner-lane nodes feed the hub ...
outof = [n_o + k for k in (3, 9, 15, 21, 27)]      # ... and the hub feeds five others (out-degree 5, in-degree 5)
spokes_in, spokes_out = [], []
for k in into:
    a, b = nodes[k], nodes[hub]
    mid = len(nodes)
    nodes.append([(a[0] + b[0]) // 2, (a[1] + b[1]) // 2])
    spokes_in += [[k, mid], [mid, hub]]
for k in outof:
    a, b = nodes[hub], nodes[k]
    mid = len(nodes)
    nodes.append([(a[0] + b[0]) // 2 + 7, (a[1] + b[1]) // 2 - 5])
    spokes_out += [[hub, mid], [mid, k]]
edges += spokes_in + spokes_out
dead = len(nodes)                                    # a dead end hanging off the outer lane (sink)
nodes += [[cx + 395, cy + 10], [cx + 440, cy + 25]]
edges += [[0, dead], [dead, dead + 1]]
edges += [[5, 5]]                                    # a self-loop (filtered by the angle list, not by the index: layer.py:122-124)
edges += [[10, 11]]                                  # a duplicate of ring edge 10 -> 11
lanepath = {"layer_color": [255, 255, 255], "nodes": nodes, "edges": edges}

# ---- lane lines
outer = ring(360, 228, 48)
inner = ring(258, 138, 40)
odd_nodes = [[300, 100], [300, 180], [300, 180], [380, 180], [380, 180], [380, 260], [520, 260], [520, 100], [700, 500],
             [410, 300], [490, 300], [450, 262], [450, 338]]
odd_edges = [[0, 1],      # vertical (layer.py:152 special case)
             [1, 2],      # zero length: two nodes on the same pixel
             [2, 3],      # horizontal
             [3, 4],      # zero length again
             [4, 5], [5, 6], [6, 7], [7, 0],
             [7, 0],      # duplicate edge
             [5, 5],      # self-loop
             [9, 10], [11, 12]]  # a cross over the hub; node 8 is isolated
single = {"layer_color": [0, 0, 255], "nodes": [[0, 0]], "edges": [[0, 0]]}  # like simple_layout's "area" layer
lanelines = {
    "outer": {"layer_color": [255, 0, 0], "nodes": outer, "edges": closed(len(outer))},
    "inner": {"layer_color": [0, 255, 0], "nodes": inner, "edges": closed(len(inner))[:-3]},   # open polyline
    "odd": {"layer_color": [255, 255, 0], "nodes": odd_nodes, "edges": odd_edges},
    "single": single,
}
# spawn points live in config_stress_graph.yaml: the spoke mid-points on both sides of the hub, the hub itself, the
# dead-end branch, the nodes before the self-loop and before the duplicated edge

with open(os.path.join(OUT, "stress_graph.json"), "w") as f:
    json.dump({"width": W, "height": H, "lanelines": lanelines, "lanepath": lanepath}, f)
succ = {}
pred = {}
for a, b in edges:
    succ.setdefault(a, []).append(b)
    pred.setdefault(b, []).append(a)
print("lanepath nodes", len(nodes), "edges", len(edges), "hub", hub, "out", len(succ[hub]), "in", len(pred[hub]),
      "sinks", [i for i in range(len(nodes)) if i not in succ])
print("spoke mid-points into the hub", [e[1] for e in spokes_in[0::2]], "out of the hub", [e[1] for e in spokes_out[0::2]])


# ---- oneway.json: a straight one-way road whose last node has two self-loops and nothing else.
# The reference RAISES on it in two documented ways (gen_golden.py `exceptions` records which states do):
#   * U-turn (maneuver 2): no lanepath edge lies within +-30 deg of the reversed heading -> local_path = [None] ->
#     TypeError at car.py:143;
#   * look-ahead reaching node 5: get_next_nodes(5) == [5, 5], both filtered from the angle list -> min() of an empty
#     range -> ValueError at layer.py:123.
ow_nodes = [[100 + 90 * i, 300] for i in range(6)]
ow_edges = [[i, i + 1] for i in range(5)] + [[5, 5], [5, 5]]
ow_lines = {"left": {"layer_color": [255, 0, 0], "nodes": [[60, 270], [640, 270]], "edges": [[0, 1]]},
            "right": {"layer_color": [0, 255, 0], "nodes": [[60, 330], [350, 330], [640, 330]], "edges": [[0, 1], [1, 2]]}}
with open(os.path.join(OUT, "oneway.json"), "w") as f:
    json.dump({"width": 700, "height": 600, "lanelines": ow_lines,
               "lanepath": {"layer_color": [255, 255, 255], "nodes": ow_nodes, "edges": ow_edges}}, f)
print("oneway: lanepath nodes", len(ow_nodes), "edges", len(ow_edges))
