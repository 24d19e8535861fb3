"""Synthetic rows x cols grid roadnets in the reference's shipped family.

Reproduces the on-disk structure of e.g.
pytsc/scenarios/cityflow/syn_2x2/2x2_roadnet.json: the
reference obtains these by shelling out to an external CityFlow checkout
(pytsc/backends/cityflow/grid_generator.py:39-42,73), which is not available,
so the family is synthesised here: 300 m spacing, 3 lanes per road
(left / straight / right), intersection width 20, the 9-entry light-phase table,
lane-link polylines = cubic Hermite curves sampled at t = 0, 0.1, ..., 1.
"""
import json

LIGHT_PHASES = [
    (5, [10, 2, 3, 6]),
    (30, [0, 2, 3, 6, 7, 10]),
    (30, [2, 3, 4, 6, 10, 11]),
    (30, [1, 2, 3, 6, 8, 10]),
    (30, [2, 3, 5, 6, 9, 10]),
    (30, [0, 1, 2, 3, 6, 10]),
    (30, [2, 3, 6, 7, 8, 10]),
    (30, [2, 3, 4, 5, 6, 10]),
    (30, [2, 3, 6, 9, 10, 11]),
]

DIRS = [(1, 0), (0, 1), (-1, 0), (0, -1)]  # 0:+x 1:+y 2:-x 3:-y


def _hermite(p0, m0, p1, m1, t):
    h00 = 2 * t**3 - 3 * t**2 + 1
    h10 = t**3 - 2 * t**2 + t
    h01 = -2 * t**3 + 3 * t**2
    h11 = t**3 - t**2
    return (
        h00 * p0[0] + h10 * m0[0] + h01 * p1[0] + h11 * m1[0],
        h00 * p0[1] + h10 * m0[1] + h01 * p1[1] + h11 * m1[1],
    )


def _lane_point(cx, cy, heading, along, lane, lane_width):
    """Point on lane `lane` of a road with `heading`, `along` metres from the
    intersection centre (cx, cy) measured along the heading."""
    dx, dy = DIRS[heading]
    rx, ry = dy, -dx  # right-hand normal
    off = (lane + 0.5) * lane_width
    return (cx + dx * along + rx * off, cy + dy * along + ry * off)


def generate_grid(rows, cols, spacing=300.0, width=20.0, n_lanes=3, lane_width=4.0,
                  max_speed=11.11):
    """Return the roadnet dict for a rows x cols signalised grid."""
    if n_lanes != 3:
        raise ValueError("the shipped family has exactly 3 lanes per road")

    def exists(x, y):
        if x < 0 or y < 0 or x > cols + 1 or y > rows + 1:
            return False
        corner = (x in (0, cols + 1)) and (y in (0, rows + 1))
        return not corner

    def virtual(x, y):
        return x in (0, cols + 1) or y in (0, rows + 1)

    def pos(x, y):
        return ((x - 1) * spacing, (y - 1) * spacing)

    def iid(x, y):
        return f"intersection_{x}_{y}"

    def has_road(x, y, d):
        nx, ny = x + DIRS[d][0], y + DIRS[d][1]
        if not exists(x, y) or not exists(nx, ny):
            return False
        return not (virtual(x, y) and virtual(nx, ny))

    def num(v):
        return int(v) if float(v).is_integer() else v

    intersections, roads = [], []
    for y in range(rows + 2):
        for x in range(cols + 2):
            if not exists(x, y):
                continue
            px, py = pos(x, y)
            for d in range(4):
                if has_road(x, y, d):
                    nx, ny = x + DIRS[d][0], y + DIRS[d][1]
                    qx, qy = pos(nx, ny)
                    roads.append(
                        {
                            "id": f"road_{x}_{y}_{d}",
                            "points": [{"x": num(px), "y": num(py)}, {"x": num(qx), "y": num(qy)}],
                            "lanes": [{"width": num(lane_width), "maxSpeed": max_speed}
                                      for _ in range(n_lanes)],
                            "startIntersection": iid(x, y),
                            "endIntersection": iid(nx, ny),
                        }
                    )
            incoming = []
            for d in range(4):  # road heading d arrives from the neighbour opposite d
                sx, sy = x - DIRS[d][0], y - DIRS[d][1]
                if has_road(sx, sy, d):
                    incoming.append((d, f"road_{sx}_{sy}_{d}"))
            outgoing = [(d, f"road_{x}_{y}_{d}") for d in range(4) if has_road(x, y, d)]
            is_virtual = virtual(x, y)
            inter = {
                "id": iid(x, y),
                "point": {"x": num(px), "y": num(py)},
                "width": 0 if is_virtual else num(width),
                "roads": [r for _, r in incoming] + [r for _, r in outgoing],
                "roadLinks": [],
                "trafficLight": None,
                "virtual": is_virtual,
            }
            if not is_virtual:
                for d, in_road in incoming:
                    for e, out_road in outgoing:
                        if e == (d + 2) % 4:
                            continue
                        if e == d:
                            typ, start_lane = "go_straight", 1
                        elif e == (d + 1) % 4:
                            typ, start_lane = "turn_left", 0
                        else:
                            typ, start_lane = "turn_right", 2
                        lane_links = []
                        p0 = _lane_point(px, py, d, -width, start_lane, lane_width)
                        m0 = (DIRS[d][0] * width, DIRS[d][1] * width)
                        m1 = (DIRS[e][0] * width, DIRS[e][1] * width)
                        for end_lane in range(n_lanes):
                            p1 = _lane_point(px, py, e, width, end_lane, lane_width)
                            pts = [_hermite(p0, m0, p1, m1, k / 10.0) for k in range(11)]
                            lane_links.append(
                                {
                                    "startLaneIndex": start_lane,
                                    "endLaneIndex": end_lane,
                                    "points": [{"x": a, "y": b} for a, b in pts],
                                }
                            )
                        inter["roadLinks"].append(
                            {
                                "type": typ,
                                "startRoad": in_road,
                                "endRoad": out_road,
                                "direction": d,
                                "laneLinks": lane_links,
                            }
                        )
                n_rl = len(inter["roadLinks"])
                inter["trafficLight"] = {
                    "roadLinkIndices": list(range(n_rl)),
                    "lightphases": [
                        {"time": t, "availableRoadLinks": list(a)} for t, a in LIGHT_PHASES
                    ],
                }
            else:
                inter["trafficLight"] = {
                    "roadLinkIndices": [],
                    "lightphases": [{"time": t, "availableRoadLinks": []} for t, _ in LIGHT_PHASES],
                }
            intersections.append(inter)
    return {"intersections": intersections, "roads": roads}


def save_grid(path, rows, cols, **kw):
    with open(path, "w") as f:
        json.dump(generate_grid(rows, cols, **kw), f)
