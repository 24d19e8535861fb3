"""Roadnet JSON -> engine geometry (lanes, lane-links, crosses, light phases).

The on-disk format is the one the reference's CityFlow backend consumes
(e.g. pytsc/scenarios/cityflow/syn_2x2/2x2_roadnet.json).
The geometry derived here is what the microsimulator (HIP kernels and the CPU
oracle alike) runs on; it restates the load-time preprocessing of the
third-party engine the reference drives through
pytsc/backends/cityflow/simulator.py:71-74.
"""
import json
import math

import numpy as np

EPS = 1e-9

LINK_TYPE = {"turn_right": 1, "turn_left": 2, "go_straight": 3}


def _sign(x):
    return 0 if abs(x) < EPS else (1 if x > 0 else -1)


def _cross_mul(ax, ay, bx, by):
    return ax * by - ay * bx


def _dot_mul(ax, ay, bx, by):
    return ax * bx + ay * by


def _seg_len(p, q):
    dx = q[0] - p[0]
    dy = q[1] - p[1]
    return math.sqrt(dx * dx + dy * dy)


def polyline_length(points):
    total = 0.0
    for i in range(len(points) - 1):
        total += _seg_len(points[i], points[i + 1])
    return total


def _on_segment(a, b, p):
    v1 = _cross_mul(b[0] - a[0], b[1] - a[1], p[0] - a[0], p[1] - a[1])
    v2 = _dot_mul(p[0] - a[0], p[1] - a[1], p[0] - b[0], p[1] - b[1])
    return _sign(v1) == 0 and _sign(v2) <= 0


def _intersect_point(a, b, c, d):
    t = _cross_mul(a[0] - c[0], a[1] - c[1], c[0] - d[0], c[1] - d[1]) / _cross_mul(
        a[0] - b[0], a[1] - b[1], c[0] - d[0], c[1] - d[1]
    )
    return (a[0] + (b[0] - a[0]) * t, a[1] + (b[1] - a[1]) * t)


def first_crossing(pa, pb):
    """First proper crossing of two polylines: (dist on a, dist on b) or None.

    Walks segment pairs in (a-major, b-minor) order, skips parallel pairs, and
    accepts the first intersection point lying on both segments.
    """
    disa = 0.0
    for ia in range(len(pa) - 1):
        a1, a2 = pa[ia], pa[ia + 1]
        disb = 0.0
        for ib in range(len(pb) - 1):
            b1, b2 = pb[ib], pb[ib + 1]
            if _sign(_cross_mul(a2[0] - a1[0], a2[1] - a1[1], b2[0] - b1[0], b2[1] - b1[1])) == 0:
                disb += _seg_len(b1, b2)
                continue
            p = _intersect_point(a1, a2, b1, b2)
            if _on_segment(a1, a2, p) and _on_segment(b1, b2, p):
                return disa + _seg_len(a1, p), disb + _seg_len(b1, p)
            disb += _seg_len(b1, b2)
        disa += _seg_len(a1, a2)
    return None


class RoadNet:
    """Flat, index-based view of a roadnet file."""

    def __init__(self, net):
        self.net = net
        self.intersections = net["intersections"]
        self.roads = net["roads"]
        self.inter_by_id = {it["id"]: it for it in self.intersections}
        self.road_by_id = {r["id"]: r for r in self.roads}
        self._build_lanes()
        self._build_links()
        self._build_crosses()

    @classmethod
    def from_file(cls, path):
        with open(path, "r") as f:
            return cls(json.load(f))

    # ------------------------------------------------------------------ lanes
    def _trimmed_road_points(self, road):
        pts = [(float(p["x"]), float(p["y"])) for p in road["points"]]
        start = self.inter_by_id[road["startIntersection"]]
        end = self.inter_by_id[road["endIntersection"]]
        if not start["virtual"]:
            w = float(start["width"])
            p1, p2 = pts[0], pts[1]
            ln = _seg_len(p1, p2)
            ux, uy = (p2[0] - p1[0]) / ln, (p2[1] - p1[1]) / ln
            pts[0] = (p1[0] + ux * w, p1[1] + uy * w)
        if not end["virtual"]:
            w = float(end["width"])
            p1, p2 = pts[-2], pts[-1]
            ln = _seg_len(p1, p2)
            ux, uy = (p2[0] - p1[0]) / ln, (p2[1] - p1[1]) / ln
            pts[-1] = (p2[0] - ux * w, p2[1] - uy * w)
        return pts

    @staticmethod
    def _offset_polyline(pts, off):
        """Lane centre line: road polyline shifted `off` to the right."""
        out = []
        n = len(pts)
        for j in range(n):
            if j == 0:
                a, b = pts[0], pts[1]
            elif j == n - 1:
                a, b = pts[n - 2], pts[n - 1]
            else:
                a, b = pts[j - 1], pts[j + 1]
            ln = _seg_len(a, b)
            ux, uy = (b[0] - a[0]) / ln, (b[1] - a[1]) / ln
            # right-hand normal of the driving direction
            vx, vy = uy, -ux
            out.append((pts[j][0] + vx * off, pts[j][1] + vy * off))
        return out

    def _build_lanes(self):
        ids = []
        for road in self.roads:
            for i in range(len(road["lanes"])):
                ids.append(f"{road['id']}_{i}")
        self.lane_ids = sorted(ids)
        self.lane_index = {lid: k for k, lid in enumerate(self.lane_ids)}
        n = len(self.lane_ids)
        self.lane_length = np.zeros(n)
        self.lane_max_speed = np.zeros(n)
        self.lane_road = [None] * n
        self.lane_idx_in_road = np.zeros(n, dtype=np.int32)
        self.lane_len_pytsc = np.zeros(n)
        self.lane_angle_deg = np.zeros(n)
        for road in self.roads:
            pts = self._trimmed_road_points(road)
            s = self.inter_by_id[road["startIntersection"]]["point"]
            e = self.inter_by_id[road["endIntersection"]]["point"]
            # reference: network_parser.py:325-352 (Euclidean centre distance)
            len_pytsc = float(
                np.linalg.norm(np.array([s["x"], s["y"]]) - np.array([e["x"], e["y"]]))
            )
            ang = math.degrees(math.atan2(e["y"] - s["y"], e["x"] - s["x"]))
            dsum = 0.0
            for i, lane in enumerate(road["lanes"]):
                k = self.lane_index[f"{road['id']}_{i}"]
                w = float(lane["width"])
                if len(pts) == 2:
                    length = _seg_len(pts[0], pts[1])
                else:
                    length = polyline_length(self._offset_polyline(pts, dsum + w / 2.0))
                dsum += w
                self.lane_length[k] = length
                self.lane_max_speed[k] = float(lane["maxSpeed"])
                self.lane_road[k] = road["id"]
                self.lane_idx_in_road[k] = i
                self.lane_len_pytsc[k] = len_pytsc
                self.lane_angle_deg[k] = ang

    # ------------------------------------------------------------------ links
    def _build_links(self):
        self.agent_ids = [it["id"] for it in self.intersections if not it["virtual"]]
        self.agent_index = {a: k for k, a in enumerate(self.agent_ids)}
        self.links = []  # dicts
        self.inconsistent = []  # road links that name undefined roads / lanes (see below)
        self.lane_out_links = [[] for _ in self.lane_ids]
        self.lane_in_links = [[] for _ in self.lane_ids]
        self.agent_links = [[] for _ in self.agent_ids]
        self.agent_n_raw_phases = []
        for it in self.intersections:
            if it["virtual"]:
                continue
            a = self.agent_index[it["id"]]
            phases = it["trafficLight"]["lightphases"]
            self.agent_n_raw_phases.append(len(phases))
            for rl_idx, rl in enumerate(it["roadLinks"]):
                mask = 0
                for p, ph in enumerate(phases):
                    if rl_idx in ph["availableRoadLinks"]:
                        mask |= 1 << p
                for ll in rl["laneLinks"]:
                    s = self.lane_index.get(f"{rl['startRoad']}_{ll['startLaneIndex']}")
                    e = self.lane_index.get(f"{rl['endRoad']}_{ll['endLaneIndex']}")
                    if s is None or e is None:
                        # a road link that names a road / lane the file does not define (the reference ships one such file,
                        # syn_1x7): the PARSER works on names, as the reference's does, and accepts it; the engine tables
                        # cannot be built from it -- pytsc_amd.tables.Tables refuses with this list (TSC_EINVAL's message)
                        self.inconsistent.append(f"intersection {it['id']}: road link {rl_idx} ({rl['startRoad']} -> {rl['endRoad']}) "
                                                 f"names lane {rl['startRoad'] if s is None else rl['endRoad']}_"
                                                 f"{ll['startLaneIndex'] if s is None else ll['endLaneIndex']}, which no road defines")
                        continue
                    pts = [(float(p["x"]), float(p["y"])) for p in ll["points"]]
                    k = len(self.links)
                    self.links.append(
                        dict(
                            start=s,
                            end=e,
                            type=LINK_TYPE[rl["type"]],
                            agent=a,
                            road_link=rl_idx,
                            mask=mask,
                            points=pts,
                            length=polyline_length(pts),
                        )
                    )
                    self.lane_out_links[s].append(k)
                    self.lane_in_links[e].append(k)
                    self.agent_links[a].append(k)

    # ---------------------------------------------------------------- crosses
    def _build_crosses(self):
        n = len(self.links)
        crosses = [[] for _ in range(n)]  # (dist_self, seq, other, dist_other)
        seq = 0
        for a_links in self.agent_links:
            m = len(a_links)
            for i in range(m):
                for j in range(i + 1, m):
                    la, lb = a_links[i], a_links[j]
                    hit = first_crossing(self.links[la]["points"], self.links[lb]["points"])
                    if hit is None:
                        continue
                    da, db = hit
                    crosses[la].append((da, seq, lb, db))
                    crosses[lb].append((db, seq, la, da))
                    seq += 1
        # stable sort by distance on this link (creation order breaks ties)
        self.link_crosses = [sorted(c, key=lambda t: (t[0], t[1])) for c in crosses]

    # ------------------------------------------------------------- conveniences
    @property
    def n_lanes(self):
        return len(self.lane_ids)

    @property
    def n_links(self):
        return len(self.links)

    @property
    def n_agents(self):
        return len(self.agent_ids)

    def lane_end_agent(self, k):
        road = self.road_by_id[self.lane_road[k]]
        return self.agent_index.get(road["endIntersection"], -1)

    def links_to_road(self, lane, road_id):
        """Lane-links leaving `lane` whose end lane belongs to `road_id`."""
        return [
            l for l in self.lane_out_links[lane] if self.lane_road[self.links[l]["end"]] == road_id
        ]

    def road_lanes(self, road_id):
        n = len(self.road_by_id[road_id]["lanes"])
        return [self.lane_index[f"{road_id}_{i}"] for i in range(n)]


def load_roadnet(config):
    """RoadNet of a Config: in-memory dict, synthetic syn_RxC grid, or file."""
    if "roadnet_json" in config.simulator:
        return RoadNet(config.simulator["roadnet_json"])
    if config.synthetic_grid is not None:
        from pytsc_amd.grid import generate_grid

        rows, cols = config.synthetic_grid
        return RoadNet(generate_grid(rows, cols))
    return RoadNet.from_file(config.roadnet_file)
