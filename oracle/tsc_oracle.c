/*
 * tsc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar f64 restatement of the hot path `TrafficSignalNetwork.step()`:
 *   - the measurement / observation / reward / mask layer follows the
 *     reference's Python (file:line cited at each function), and is pinned by
 *     the golden fixtures under tests/golden/ that were produced by importing
 *     the reference's own Python layer on top of this engine
 *     (tests/golden/gen_golden.py);
 *   - the engine sub-step restates the published algorithm of the third-party
 *     dependency CityFlow (cityflow-project/CityFlow; PyTSC pins no version,
 *     it is absent from /root/reference and cannot be built here), anchored on
 *     the reference's call sites pytsc/backends/cityflow/simulator.py:71-95,
 *     retriever.py:35,94-112, traffic_signal.py:31,58.  ENGINE PARITY WITH
 *     CITYFLOW IS UNPINNED: no reference test, fixture or recorded trace pins
 *     it; only analytic known-answer tests do (tests/test_engine_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It walks the data the obvious way (per-vehicle structs,
 * per-drivable id lists, the engine's own phase order: spawn -> notifyCross ->
 * getAction -> updateLocation -> updateAction -> updateLeaderAndGap) and is
 * deliberately structured differently from the HIP kernels.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pytsc_amd.h"

#define ST_NONE 0
#define ST_WAITING 1
#define ST_RUNNING 2
#define ST_FINISHED 3

typedef struct {
  int state;
  int route, tmpl, seq;   /* seq: index into the route's drivable sequence */
  double dist, speed;
  int enter_ll_time;
  int blocker;            /* vid or -1, committed */
  int leader;             /* vid or -1 (updateLeaderAndGap) */
  double gap;
  int enter_time;         /* creation step */
  /* buffer (Vehicle::buffer) */
  double b_dist, b_speed;
  int b_seq, b_end, b_drv_set, b_blocker_set, b_blocker;
  int b_ell_set, b_ell;
  int is_new;             /* inserted by handleWaiting of the current step */
} veh_t;

typedef struct {
  int n;
  int *v; /* vids front (index 0) to back */
} dlist_t;

typedef struct {
  int set;            /* flow set in use */
  int step;           /* engine step counter */
  int n_veh;          /* vehicles in this set */
  veh_t *veh;
  dlist_t *drv;       /* [n_drv] */
  int *spawn_cur;     /* [n_lanes] cursor into the lane's waiting order */
  int *raw_phase;     /* [n_agents] */
  int *cur_idx, *time_on_phase, *phase_changed; /* [n_agents] */
  int finished_cnt;
  long long finished_tsum; /* sum over finished vehicles of (finish_time - enter_time) */
  int *notify_veh;    /* [n_cross_entries] */
  double *notify_dist;
  int restarts;
  /* last measurement */
  int *lane_n, *lane_q;
  double *lane_sum_speed, *lane_posmat; /* [n_lanes*max_bins] */
  int have_phase_changed;
} rep_t;

typedef struct {
  tsc_net_t net;
  tsc_flows_t fl;
  tsc_config_t cfg;
  int n_drv, n_cross;
  int *cross_peer;   /* entry index of the same cross in the other link's list */
  int max_bins;
  int errors;
  rep_t *rep;
} orc_t;

/* ------------------------------------------------------------------ helpers */
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max2(double a, double b) { return a > b ? a : b; }
static inline const double *T(const orc_t *o, int tmpl) { return o->net.tmpl + (size_t)tmpl * TSC_TMPL_STRIDE; }
static inline int is_link(const orc_t *o, int d) { return d >= o->net.n_lanes; }
static inline int seq_drv(const orc_t *o, const veh_t *v, int seq) {
  int b = o->net.route_ptr[v->route], e = o->net.route_ptr[v->route + 1];
  if (seq < 0 || b + seq >= e) return -1;
  return o->net.route_drv[b + seq];
}
static inline int cur_drv(const orc_t *o, const veh_t *v) { return seq_drv(o, v, v->seq); }
/* Router::getNextDrivable(i) relative to the committed drivable */
static inline int next_drv(const orc_t *o, const veh_t *v, int i) { return seq_drv(o, v, v->seq + 1 + i); }

/* C++ (int)x on x86-64 (cvttsd2si): out-of-range -> INT_MIN */
static inline int cvt_int(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return INT_MIN;
  return (int)x;
}

/* std::to_string(double) -> float(): value rounded to 6 decimals
 * (retriever.py:35-44 reads get_vehicle_info strings) */
static double round6(double x) {
  double k = rint(x * 1e6);
  double r = fma(x, 1e6, -k);
  if (r > 0.5) k += 1.0;
  else if (r < -0.5) k -= 1.0;
  return k / 1e6;
}

/* CPython float floor division a // b for b > 0 */
static double py_floordiv(double a, double b) {
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && (mod < 0.0)) { div -= 1.0; }
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

/* numpy's pairwise summation for a contiguous double array (add.reduce) */
static double np_pairwise_sum(const double *a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}

/* ------------------------------------------------------- vehicle kinematics */
static double min_brake_distance(const orc_t *o, const veh_t *v) {
  return 0.5 * v->speed * v->speed / T(o, v->tmpl)[TSC_T_MAXNEGACC];
}
static int can_yield(const orc_t *o, const veh_t *v, double dist) {
  const double *t = T(o, v->tmpl);
  return (dist > 0 && min_brake_distance(o, v) < dist - t[TSC_T_YIELDDIST]) ||
         (dist < 0 && dist + t[TSC_T_LEN] < 0);
}
static double distance_until_speed(const orc_t *o, const veh_t *v, double speed, double acc) {
  if (speed <= v->speed) return 0;
  double interval = o->cfg.interval;
  int stage1steps = (int)floor((speed - v->speed) / acc / interval);
  double stage1speed = v->speed + stage1steps * acc / interval;
  double stage1dis = (v->speed + stage1speed) * (stage1steps * interval) / 2;
  return stage1dis + (stage1speed < speed ? ((stage1speed + speed) * interval / 2) : 0);
}
static int reach_steps(const orc_t *o, const veh_t *v, double distance, double target, double acc) {
  if (distance <= 0) return 0;
  if (v->speed > target) return (int)ceil(distance / v->speed);
  double dut = distance_until_speed(o, v, target, acc);
  double interval = o->cfg.interval;
  if (dut > distance) {
    return (int)ceil((sqrt(v->speed * v->speed + 2 * acc * distance) - v->speed) / acc / interval);
  } else {
    return (int)ceil((target - v->speed) / acc / interval) +
           (int)ceil((distance - dut) / target / interval);
  }
}
static int reach_steps_on_link(const orc_t *o, const veh_t *v, double distance, int link) {
  const double *t = T(o, v->tmpl);
  int type = o->net.link_type[link - o->net.n_lanes];
  int turn = type != TSC_LINK_STRAIGHT;
  return reach_steps(o, v, distance, turn ? t[TSC_T_TURNSPEED] : t[TSC_T_MAXSPEED], t[TSC_T_MAXPOSACC]);
}
static double no_collision_speed(double vL, double dL, double vF, double dF, double gap,
                                 double interval, double targetGap) {
  double c = vF * interval / 2 + targetGap - 0.5 * vL * vL / dL - gap;
  double a = 0.5 / dF;
  double b = 0.5 * interval;
  if (b * b < 4 * a * c) return -100;
  double v1 = 0.5 / a * (sqrt(b * b - 4 * a * c) - b);
  double v2 = 2 * vL - dL * interval + 2 * (gap - targetGap) / interval;
  return min2(v1, v2);
}
static double car_follow_speed(const orc_t *o, const rep_t *r, const veh_t *v) {
  const double *t = T(o, v->tmpl);
  double interval = o->cfg.interval;
  if (v->leader < 0) return t[TSC_T_MAXSPEED];
  const veh_t *L = &r->veh[v->leader];
  const double *tl = T(o, L->tmpl);
  double s = no_collision_speed(L->speed, tl[TSC_T_MAXNEGACC], v->speed, t[TSC_T_MAXNEGACC], v->gap, interval, 0);
  double assumeDecel = 0, leaderSpeed = L->speed;
  if (v->speed > leaderSpeed) assumeDecel = v->speed - leaderSpeed;
  s = min2(s, no_collision_speed(L->speed, tl[TSC_T_USUALNEGACC], v->speed, t[TSC_T_USUALNEGACC], v->gap,
                                 interval, t[TSC_T_MINGAP]));
  s = min2(s, (v->gap + (leaderSpeed + assumeDecel / 2) * interval - v->speed * interval / 2) /
                  (t[TSC_T_HEADWAY] + interval / 2));
  return s;
}
static double stop_before_speed(const orc_t *o, const veh_t *v, double distance) {
  const double *t = T(o, v->tmpl);
  double interval = o->cfg.interval;
  /* getBrakeDistanceAfterAccel(maxPosAcc, usualNegAcc, interval) */
  double nextSpeed = v->speed + t[TSC_T_MAXPOSACC] * interval;
  double bd = (v->speed + nextSpeed) * interval / 2 + (nextSpeed * nextSpeed / t[TSC_T_USUALNEGACC] / 2);
  if (bd < distance) return v->speed + t[TSC_T_MAXPOSACC] * interval;
  double takeInterval = 2 * distance / (v->speed + 1e-9) / interval;
  if (takeInterval >= 1) {
    return v->speed - v->speed / cvt_int(takeInterval);
  } else {
    return v->speed - v->speed / takeInterval;
  }
}

/* ------------------------------------------------------------ lane helpers */
static inline int last_vid(const rep_t *r, int d) { return r->drv[d].n ? r->drv[d].v[r->drv[d].n - 1] : -1; }
static inline int first_vid(const rep_t *r, int d) { return r->drv[d].n ? r->drv[d].v[0] : -1; }

static int link_available(const orc_t *o, const rep_t *r, int link) {
  int k = link - o->net.n_lanes;
  return (o->net.link_avail_mask[k] >> r->raw_phase[o->net.link_agent[k]]) & 1u;
}
static int lane_can_enter(const orc_t *o, const rep_t *r, int lane, const veh_t *v) {
  int tv = last_vid(r, lane);
  if (tv < 0) return 1;
  const veh_t *tail = &r->veh[tv];
  return tail->dist > T(o, tail->tmpl)[TSC_T_LEN] + T(o, v->tmpl)[TSC_T_LEN] || tail->speed >= 2;
}
static int lane_available(const orc_t *o, const rep_t *r, int lane, const veh_t *v) {
  int tv = last_vid(r, lane);
  if (tv < 0) return 1;
  const veh_t *tail = &r->veh[tv];
  return tail->dist > T(o, tail->tmpl)[TSC_T_LEN] + T(o, v->tmpl)[TSC_T_MINGAP];
}

/* Vehicle::updateLeaderAndGap(leader) */
static void update_leader_and_gap(const orc_t *o, rep_t *r, veh_t *v, int leader_vid) {
  int cd = cur_drv(o, v);
  if (leader_vid >= 0 && cur_drv(o, &r->veh[leader_vid]) == cd) {
    const veh_t *L = &r->veh[leader_vid];
    v->leader = leader_vid;
    v->gap = L->dist - T(o, L->tmpl)[TSC_T_LEN] - v->dist;
    return;
  }
  v->leader = -1;
  const double *t = T(o, v->tmpl);
  double dis = o->net.drv_length[cd] - v->dist;
  for (int i = 0;; ++i) {
    int d = next_drv(o, v, i);
    if (d < 0) return;
    if (is_link(o, d)) {
      int sl = o->net.link_start_lane[d - o->net.n_lanes];
      for (int k = o->net.lane_out_ptr[sl]; k < o->net.lane_out_ptr[sl + 1]; k++) {
        int ll = o->net.lane_out_links[k];
        int cv = last_vid(r, ll);
        if (cv >= 0) {
          const veh_t *cand = &r->veh[cv];
          double cg = dis + cand->dist - T(o, cand->tmpl)[TSC_T_LEN];
          if (v->leader < 0 || cg < v->gap) {
            v->leader = cv;
            v->gap = cg;
          }
        }
      }
      if (v->leader >= 0) return;
    } else {
      int cv = last_vid(r, d);
      if (cv >= 0) {
        const veh_t *cand = &r->veh[cv];
        v->leader = cv;
        v->gap = dis + cand->dist - T(o, cand->tmpl)[TSC_T_LEN];
        return;
      }
    }
    dis += o->net.drv_length[d];
    if (dis > t[TSC_T_MAXSPEED] * t[TSC_T_MAXSPEED] / t[TSC_T_USUALNEGACC] / 2 +
                  t[TSC_T_MAXSPEED] * o->cfg.interval * 2)
      return;
  }
}

/* ------------------------------------------------------------------ crosses */
static void notify_cross(const orc_t *o, rep_t *r) {
  const tsc_net_t *n = &o->net;
  for (int e = 0; e < o->n_cross; e++) r->notify_veh[e] = -1;
  for (int k = 0; k < n->n_links; k++) {
    int link = n->n_lanes + k;
    int cb = n->link_cross_ptr[k], ce = n->link_cross_ptr[k + 1];
    int ri = ce - 1; /* reverse iterator over this link's crosses */
    double L = n->drv_length[link];
    /* first check the vehicle on the end lane */
    int ev = last_vid(r, n->link_end_lane[k]);
    if (ev >= 0) {
      const veh_t *v = &r->veh[ev];
      if (seq_drv(o, v, v->seq - 1) == link) {
        double vehDistance = v->dist - T(o, v->tmpl)[TSC_T_LEN];
        while (ri >= cb) {
          double crossDistance = L - n->cross_dist_self[ri];
          if (crossDistance + vehDistance < 0 /* leaveDistance */) {
            r->notify_veh[ri] = ev;
            r->notify_dist[ri] = -(v->dist + crossDistance);
            --ri;
          } else break;
        }
      }
    }
    /* check each vehicle on the lane-link */
    for (int i = 0; i < r->drv[link].n; i++) {
      int lv = r->drv[link].v[i];
      const veh_t *v = &r->veh[lv];
      double vehDistance = v->dist;
      while (ri >= cb) {
        double crossDistance = n->cross_dist_self[ri];
        if (vehDistance > crossDistance) {
          if (vehDistance - crossDistance - T(o, v->tmpl)[TSC_T_LEN] <= 0 /* leaveDistance */) {
            r->notify_veh[ri] = lv;
            r->notify_dist[ri] = crossDistance - vehDistance;
          } else break;
        } else {
          r->notify_veh[ri] = lv;
          r->notify_dist[ri] = crossDistance - vehDistance;
        }
        --ri;
      }
    }
    /* check the first vehicle on the incoming lane */
    int sl = n->link_start_lane[k];
    int sv = first_vid(r, sl);
    if (sv >= 0) {
      const veh_t *v = &r->veh[sv];
      if (next_drv(o, v, 0) == link && link_available(o, r, link)) {
        double vehDistance = n->drv_length[sl] - v->dist;
        while (ri >= cb) {
          r->notify_veh[ri] = sv;
          r->notify_dist[ri] = vehDistance + n->cross_dist_self[ri];
          --ri;
        }
      }
    }
  }
}

/* Cross::canPass for `v` on/approaching `link` at this link's cross entry `e` */
static int cross_can_pass(const orc_t *o, const rep_t *r, int vid, int link, int e,
                          double distanceToLaneLinkStart) {
  const tsc_net_t *n = &o->net;
  const veh_t *v = &r->veh[vid];
  int pe = o->cross_peer[e];
  int fv = r->notify_veh[pe];
  if (fv < 0) return 1;
  const veh_t *foe = &r->veh[fv];
  int other = n->cross_other[e];
  int t1 = n->link_type[link - n->n_lanes];
  int t2 = n->link_type[other - n->n_lanes];
  double d1 = n->cross_dist_self[e] - distanceToLaneLinkStart;
  double d2 = r->notify_dist[pe];
  if (!can_yield(o, v, d1)) return 1;
  int yield = 0;
  if (!can_yield(o, foe, d2)) yield = 1;
  if (yield == 0) {
    if (t1 > t2) {
      yield = -1;
    } else if (t1 < t2) {
      if (d2 > 0) {
        int foeSteps = reach_steps_on_link(o, foe, d2, other);
        int mySteps = reach_steps_on_link(o, v, d1, link);
        if (foeSteps > mySteps) yield = -1;
      } else {
        if (d2 + T(o, foe->tmpl)[TSC_T_LEN] < 0) yield = -1;
      }
      if (yield == 0) yield = 1;
    } else {
      if (d2 > 0) {
        int foeSteps = reach_steps_on_link(o, foe, d2, other);
        int mySteps = reach_steps_on_link(o, v, d1, link);
        if (foeSteps > mySteps) {
          yield = -1;
        } else if (foeSteps < mySteps) {
          yield = 1;
        } else {
          if (v->enter_ll_time == foe->enter_ll_time) {
            if (d1 == d2) {
              /* engine: the vehicles' random priorities (std::mt19937 of the config seed, drawn at creation); the
               * tables either carry a restatement of that stream (tsc_flows_t::veh_priority: higher goes first) or
               * none: lower vid goes first */
              if (o->fl.veh_priority) {
                const int base = o->fl.set_veh_ptr[r->set];
                const int p1 = o->fl.veh_priority[base + vid], p2 = o->fl.veh_priority[base + fv];
                yield = p1 != p2 ? (p1 > p2 ? -1 : 1) : (vid < fv ? -1 : 1);
              } else {
                yield = vid < fv ? -1 : 1;
              }
            } else {
              yield = d1 < d2 ? -1 : 1;
            }
          } else {
            yield = v->enter_ll_time < foe->enter_ll_time ? -1 : 1;
          }
        }
      } else {
        yield = d2 + T(o, foe->tmpl)[TSC_T_LEN] < 0 ? -1 : 1;
      }
    }
  }
  if (yield == 1) {
    /* blocker-chain cycle check (Floyd) */
    int fast = fv, slow = fv;
    while (fast >= 0 && r->veh[fast].blocker >= 0) {
      slow = r->veh[slow].blocker;
      fast = r->veh[r->veh[fast].blocker].blocker;
      if (slow == fast) {
        yield = -1;
        break;
      }
    }
  }
  return yield == -1;
}

static double intersection_related_speed(const orc_t *o, rep_t *r, int vid) {
  const tsc_net_t *n = &o->net;
  veh_t *v = &r->veh[vid];
  const double *t = T(o, v->tmpl);
  double s = t[TSC_T_MAXSPEED];
  int cd = cur_drv(o, v);
  int nd = next_drv(o, v, 0);
  int link = -1;
  if (nd >= 0 && is_link(o, nd)) {
    link = nd;
    int k = link - n->n_lanes;
    if (!link_available(o, r, link) || !lane_can_enter(o, r, n->link_end_lane[k], v)) {
      if (min_brake_distance(o, v) > n->drv_length[cd] - v->dist) {
        /* cannot brake before the stop line: keep going */
      } else {
        s = min2(s, stop_before_speed(o, v, n->drv_length[cd] - v->dist));
        return s;
      }
    }
    if (n->link_type[k] != TSC_LINK_STRAIGHT) s = min2(s, t[TSC_T_TURNSPEED]);
  }
  if (link < 0 && is_link(o, cd)) link = cd;
  double distanceToLaneLinkStart = !is_link(o, cd) ? -(n->drv_length[cd] - v->dist) : v->dist;
  int k = link - n->n_lanes;
  for (int e = n->link_cross_ptr[k]; e < n->link_cross_ptr[k + 1]; e++) {
    double distanceOnLaneLink = n->cross_dist_self[e];
    if (distanceOnLaneLink < distanceToLaneLinkStart) continue;
    if (!cross_can_pass(o, r, vid, link, e, distanceToLaneLinkStart)) {
      s = min2(s, stop_before_speed(o, v, distanceOnLaneLink - distanceToLaneLinkStart - t[TSC_T_YIELDDIST]));
      v->b_blocker_set = 1;
      v->b_blocker = r->notify_veh[o->cross_peer[e]];
      break;
    }
  }
  return s;
}

static int is_intersection_related(const orc_t *o, const veh_t *v) {
  int cd = cur_drv(o, v);
  if (is_link(o, cd)) return 1;
  int nd = next_drv(o, v, 0);
  if (nd >= 0 && is_link(o, nd) &&
      o->net.drv_length[cd] - v->dist <= T(o, v->tmpl)[TSC_T_APPROACH])
    return 1;
  return 0;
}

/* Engine::vehicleControl */
static void vehicle_control(const orc_t *o, rep_t *r, int vid) {
  veh_t *v = &r->veh[vid];
  const double *t = T(o, v->tmpl);
  double interval = o->cfg.interval;
  int cd = cur_drv(o, v);
  double s = t[TSC_T_MAXSPEED];
  s = min2(s, v->speed + t[TSC_T_MAXPOSACC] * interval);
  s = min2(s, o->net.drv_max_speed[cd]);
  s = min2(s, car_follow_speed(o, r, v));
  if (is_intersection_related(o, v)) s = min2(s, intersection_related_speed(o, r, vid));
  s = max2(s, v->speed - t[TSC_T_MAXNEGACC] * interval);
  double nextSpeed = s, deltaDis, speed = v->speed;
  if (nextSpeed < 0) {
    deltaDis = 0.5 * speed * speed / t[TSC_T_MAXNEGACC];
    nextSpeed = 0;
  } else {
    deltaDis = (speed + nextSpeed) * interval / 2;
  }
  v->b_speed = nextSpeed;
  /* Vehicle::setDeltaDistance */
  double dis = deltaDis + v->dist;
  int seq = v->seq;
  int d = cd;
  v->b_end = 0;
  v->b_drv_set = 0;
  for (int i = 0; d >= 0 && dis > o->net.drv_length[d]; ++i) {
    dis -= o->net.drv_length[d];
    int nd = next_drv(o, v, i);
    if (nd < 0) v->b_end = 1;
    d = nd;
    seq = v->seq + 1 + i;
    v->b_drv_set = 1;
  }
  v->b_seq = seq;
  v->b_dist = dis;
}

/* ------------------------------------------------------------------ sub-step */
static void list_push(orc_t *o, rep_t *r, int d, int vid) {
  dlist_t *l = &r->drv[d];
  if (l->n >= o->net.drv_cap[d]) {
    o->errors++;
    int *nv = (int *)realloc(l->v, sizeof(int) * (size_t)(l->n + 16));
    l->v = nv; /* oracle keeps going; the error counter records the overflow */
  }
  l->v[l->n++] = vid;
}

static void handle_waiting(orc_t *o, rep_t *r) {
  const tsc_net_t *n = &o->net;
  const tsc_flows_t *f = &o->fl;
  int base = f->set_veh_ptr[r->set];
  for (int lane = 0; lane < n->n_lanes; lane++) {
    int sb = f->spawn_ptr[r->set * n->n_lanes + lane], se = f->spawn_ptr[r->set * n->n_lanes + lane + 1];
    int c = r->spawn_cur[lane];
    if (sb + c >= se) continue;
    int vid = f->spawn_vid[sb + c];
    if (f->veh_start[base + vid] > r->step) continue; /* not created yet */
    veh_t *v = &r->veh[vid];
    if (v->state == ST_NONE) { /* Flow::nextStep + planRoute */
      v->state = ST_WAITING;
    }
    if (lane_available(o, r, lane, v)) {
      int tail = last_vid(r, lane);
      v->state = ST_RUNNING;
      v->dist = 0;
      v->speed = 0;
      v->seq = 0;
      v->enter_ll_time = INT_MAX;
      v->blocker = -1;
      v->is_new = 1;
      list_push(o, r, lane, vid);
      update_leader_and_gap(o, r, v, tail);
      r->spawn_cur[lane] = c + 1;
    }
  }
}

typedef struct { int vid; double dis; } push_t;
static int push_cmp(const void *a, const void *b) {
  const push_t *x = (const push_t *)a, *y = (const push_t *)b;
  if (x->dis > y->dis) return -1;
  if (x->dis < y->dis) return 1;
  return x->vid < y->vid ? -1 : (x->vid > y->vid ? 1 : 0); /* deterministic tie-break */
}

static void engine_step(orc_t *o, rep_t *r) {
  const tsc_net_t *n = &o->net;
  handle_waiting(o, r);
  notify_cross(o, r);
  /* getAction: all decisions read the committed state (Jacobi) */
  int n_push = 0;
  static __thread push_t *pushbuf = NULL;
  static __thread int pushcap = 0;
  for (int d = 0; d < o->n_drv; d++) {
    for (int i = 0; i < r->drv[d].n; i++) {
      int vid = r->drv[d].v[i];
      veh_t *v = &r->veh[vid];
      v->b_blocker_set = 0;
      v->b_ell_set = 0;
      vehicle_control(o, r, vid);
      if (!v->b_end && v->b_drv_set) {
        if (n_push >= pushcap) {
          pushcap = pushcap ? pushcap * 2 : 256;
          pushbuf = (push_t *)realloc(pushbuf, sizeof(push_t) * (size_t)pushcap);
        }
        pushbuf[n_push].vid = vid;
        pushbuf[n_push].dis = v->b_dist;
        n_push++;
      }
    }
  }
  /* updateLocation */
  for (int d = 0; d < o->n_drv; d++) {
    dlist_t *l = &r->drv[d];
    int w = 0;
    for (int i = 0; i < l->n; i++) {
      int vid = l->v[i];
      veh_t *v = &r->veh[vid];
      if (v->b_drv_set || v->b_end) {
        if (v->b_end) {
          v->state = ST_FINISHED; /* removed from the pool */
          r->finished_tsum += (long long)r->step - v->enter_time;
          r->finished_cnt += 1;
        }
      } else {
        l->v[w++] = vid;
      }
    }
    l->n = w;
  }
  qsort(pushbuf, (size_t)n_push, sizeof(push_t), push_cmp);
  for (int i = 0; i < n_push; i++) {
    veh_t *v = &r->veh[pushbuf[i].vid];
    int d = seq_drv(o, v, v->b_seq);
    list_push(o, r, d, pushbuf[i].vid);
    v->b_ell_set = 1;
    v->b_ell = is_link(o, d) ? r->step : INT_MAX;
  }
  /* updateAction */
  for (int vid = 0; vid < r->n_veh; vid++) {
    veh_t *v = &r->veh[vid];
    if (v->state != ST_RUNNING) continue;
    if (v->b_blocker_set && r->veh[v->b_blocker].state == ST_FINISHED) {
      /* blocker left the network during this step */
      v->b_blocker_set = 0;
    }
    v->dist = v->b_dist;
    v->speed = v->b_speed;
    v->seq = v->b_seq;
    if (v->b_ell_set) v->enter_ll_time = v->b_ell;
    v->blocker = v->b_blocker_set ? v->b_blocker : -1;
    v->is_new = 0;
  }
  /* updateLeaderAndGap */
  for (int d = 0; d < o->n_drv; d++) {
    int leader = -1;
    for (int i = 0; i < r->drv[d].n; i++) {
      int vid = r->drv[d].v[i];
      update_leader_and_gap(o, r, &r->veh[vid], leader);
      leader = vid;
    }
  }
  (void)n;
  r->step += 1;
}

/* ---------------------------------------------------------------- lifecycle */
static void rep_reset(orc_t *o, rep_t *r, int rep_index, int advance_set, int forced_set) {
  const tsc_net_t *n = &o->net;
  if (advance_set) {
    if (o->cfg.sequential_sets) r->restarts += 1;
  }
  r->set = (o->cfg.set_offset + rep_index + r->restarts) % o->fl.n_sets;
  /* flow_rate_type "random" (backends/cityflow/config.py:63-76): the caller's draw */
  if (forced_set >= 0) r->set = forced_set % o->fl.n_sets;
  r->n_veh = o->fl.set_veh_ptr[r->set + 1] - o->fl.set_veh_ptr[r->set];
  r->step = 0;
  int base = o->fl.set_veh_ptr[r->set];
  for (int i = 0; i < r->n_veh; i++) {
    veh_t *v = &r->veh[i];
    memset(v, 0, sizeof(*v));
    v->route = o->fl.veh_route[base + i];
    v->tmpl = o->fl.veh_tmpl[base + i];
    v->enter_time = o->fl.veh_start[base + i];
    v->leader = -1;
    v->blocker = -1;
    v->enter_ll_time = INT_MAX;
  }
  for (int d = 0; d < o->n_drv; d++) r->drv[d].n = 0;
  memset(r->spawn_cur, 0, sizeof(int) * (size_t)n->n_lanes);
  r->finished_cnt = 0;
  r->finished_tsum = 0;
  r->have_phase_changed = 0;
  for (int a = 0; a < n->n_agents; a++) {
    /* TSProgram start phase (backends/cityflow/traffic_signal.py:20-32,
     * common/traffic_signal.py:83-92) */
    r->cur_idx[a] = 0;
    r->time_on_phase[a] = 0;
    r->phase_changed[a] = 0;
    r->raw_phase[a] = n->agent_phase_raw[a * TSC_MAX_AGENT_PHASES + 0];
  }
}

void *tsco_create(const tsc_net_t *net, const tsc_flows_t *fl, const tsc_config_t *cfg) {
  orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
  o->net = *net;
  o->fl = *fl;
  o->cfg = *cfg;
  o->n_drv = net->n_lanes + net->n_links;
  o->n_cross = net->link_cross_ptr[net->n_links];
  o->cross_peer = (int *)malloc(sizeof(int) * (size_t)(o->n_cross + 1));
  for (int k = 0; k < net->n_links; k++) {
    for (int e = net->link_cross_ptr[k]; e < net->link_cross_ptr[k + 1]; e++) {
      int other = net->cross_other[e] - net->n_lanes;
      int peer = -1;
      for (int e2 = net->link_cross_ptr[other]; e2 < net->link_cross_ptr[other + 1]; e2++) {
        if (net->cross_other[e2] == net->n_lanes + k && net->cross_dist_self[e2] == net->cross_dist_other[e] &&
            net->cross_dist_other[e2] == net->cross_dist_self[e]) {
          peer = e2;
          break;
        }
      }
      if (peer < 0) o->errors++;
      o->cross_peer[e] = peer;
    }
  }
  int max_bins = cfg->visibility;
  for (int l = 0; l < net->n_lanes; l++)
    if (net->lane_bins[l] > max_bins) max_bins = net->lane_bins[l];
  o->max_bins = max_bins;
  int max_veh = 1;
  for (int s = 0; s < fl->n_sets; s++) {
    int nv = fl->set_veh_ptr[s + 1] - fl->set_veh_ptr[s];
    if (nv > max_veh) max_veh = nv;
  }
  o->rep = (rep_t *)calloc((size_t)cfg->n_replicas, sizeof(rep_t));
  for (int i = 0; i < cfg->n_replicas; i++) {
    rep_t *r = &o->rep[i];
    r->veh = (veh_t *)calloc((size_t)max_veh, sizeof(veh_t));
    r->drv = (dlist_t *)calloc((size_t)o->n_drv, sizeof(dlist_t));
    for (int d = 0; d < o->n_drv; d++) r->drv[d].v = (int *)malloc(sizeof(int) * (size_t)(net->drv_cap[d] + 1));
    r->spawn_cur = (int *)calloc((size_t)net->n_lanes, sizeof(int));
    r->raw_phase = (int *)calloc((size_t)net->n_agents * 4, sizeof(int));
    r->cur_idx = r->raw_phase + net->n_agents;
    r->time_on_phase = r->cur_idx + net->n_agents;
    r->phase_changed = r->time_on_phase + net->n_agents;
    r->notify_veh = (int *)malloc(sizeof(int) * (size_t)(o->n_cross + 1));
    r->notify_dist = (double *)malloc(sizeof(double) * (size_t)(o->n_cross + 1));
    r->lane_n = (int *)calloc((size_t)net->n_lanes * 2, sizeof(int));
    r->lane_q = r->lane_n + net->n_lanes;
    r->lane_sum_speed = (double *)calloc((size_t)net->n_lanes, sizeof(double));
    r->lane_posmat = (double *)calloc((size_t)net->n_lanes * (size_t)max_bins, sizeof(double));
    r->restarts = 0;
    rep_reset(o, r, i, 0, -1);
  }
  return o;
}

void tsco_destroy(void *h) {
  orc_t *o = (orc_t *)h;
  for (int i = 0; i < o->cfg.n_replicas; i++) {
    rep_t *r = &o->rep[i];
    for (int d = 0; d < o->n_drv; d++) free(r->drv[d].v);
    free(r->veh); free(r->drv); free(r->spawn_cur); free(r->raw_phase);
    free(r->notify_veh); free(r->notify_dist); free(r->lane_n);
    free(r->lane_sum_speed); free(r->lane_posmat);
  }
  free(o->rep);
  free(o->cross_peer);
  free(o);
}

int tsco_errors(void *h) { return ((orc_t *)h)->errors; }

/* --- engine-level API (what a cityflow.Engine-compatible shim needs) ------ */
void tsco_engine_next_step(void *h, int rep) { engine_step((orc_t *)h, &((orc_t *)h)->rep[rep]); }
void tsco_engine_set_tl_phase(void *h, int rep, int agent, int raw) { ((orc_t *)h)->rep[rep].raw_phase[agent] = raw; }
double tsco_engine_current_time(void *h, int rep) { return ((orc_t *)h)->rep[rep].step * ((orc_t *)h)->cfg.interval; }
int tsco_engine_vehicle_count(void *h, int rep) {
  orc_t *o = (orc_t *)h;
  rep_t *r = &o->rep[rep];
  int c = 0;
  for (int i = 0; i < r->n_veh; i++) c += r->veh[i].state == ST_RUNNING;
  return c;
}
double tsco_engine_average_travel_time(void *h, int rep) {
  orc_t *o = (orc_t *)h;
  rep_t *r = &o->rep[rep];
  /* Engine::getAverageTravelTime: finished durations + (now - enterTime) of
   * every pooled (created, unfinished, waiting included) vehicle */
  double now = r->step * o->cfg.interval;
  double tt = (double)r->finished_tsum * o->cfg.interval;
  int n = r->finished_cnt;
  int base = o->fl.set_veh_ptr[r->set];
  for (int i = 0; i < r->n_veh; i++) {
    if (r->veh[i].state == ST_FINISHED) continue;
    /* created at the start of step s iff veh_start <= s, for s < r->step */
    if (o->fl.veh_start[base + i] < r->step) {
      tt += now - o->fl.veh_start[base + i] * o->cfg.interval;
      n++;
    }
  }
  return n == 0 ? 0 : tt / n;
}
/* vehicles of a lane, front to back; returns count */
int tsco_engine_lane_vehicles(void *h, int rep, int lane, int cap, int *vids, double *dist, double *speed) {
  orc_t *o = (orc_t *)h;
  rep_t *r = &o->rep[rep];
  int n = r->drv[lane].n;
  for (int i = 0; i < n && i < cap; i++) {
    const veh_t *v = &r->veh[r->drv[lane].v[i]];
    vids[i] = r->drv[lane].v[i];
    dist[i] = v->dist;
    speed[i] = v->speed;
  }
  return n;
}
int tsco_engine_drv_count(void *h, int rep, int d) { return ((orc_t *)h)->rep[rep].drv[d].n; }
int tsco_engine_finished(void *h, int rep) { return ((orc_t *)h)->rep[rep].finished_cnt; }
int tsco_engine_created(void *h, int rep) {
  orc_t *o = (orc_t *)h;
  rep_t *r = &o->rep[rep];
  int t = r->step < o->fl.sim_len + 1 ? r->step : o->fl.sim_len + 1;
  return o->fl.created_cnt[r->set * (o->fl.sim_len + 2) + t];
}

/* --------------------------------------------------------------- measurement */
/* retriever.py:54-85 (_compute_lane_measurements) + :20-52 (position matrix) */
static void measure_lanes(orc_t *o, rep_t *r) {
  const tsc_net_t *n = &o->net;
  int vis = o->cfg.visibility;
  for (int l = 0; l < n->n_lanes; l++) {
    int cnt = r->drv[l].n, q = 0;
    double sum = 0;
    double *pm = r->lane_posmat + (size_t)l * o->max_bins;
    for (int b = 0; b < o->max_bins; b++) pm[b] = -1.0;
    int bins = n->lane_bins[l];
    double len = n->lane_len_pytsc[l];
    for (int i = 0; i < cnt; i++) {
      const veh_t *v = &r->veh[r->drv[l].v[i]];
      if (v->speed < 0.1) q++;
      sum += v->speed;
      if (bins > 0) {
        double pos = round6(v->dist);
        if (pos < 0) pos = 0;
        else if (pos > len) pos = len;
        double bin_size = len / bins;
        int idx = (int)py_floordiv(pos, bin_size);
        if (idx >= bins) idx = bins - 1;
        double norm_speed = round6(v->speed) / n->drv_max_speed[l];
        pm[idx] += 1.0;
        pm[idx] += norm_speed;
      }
    }
    (void)vis;
    r->lane_n[l] = cnt;
    r->lane_q[l] = q;
    r->lane_sum_speed[l] = sum;
  }
}

static inline double lane_mean_speed(const rep_t *r, int l) {
  return r->lane_n[l] == 0 ? 0.0 : r->lane_sum_speed[l] / r->lane_n[l];
}
static inline double lane_occupancy(const orc_t *o, const rep_t *r, int l) {
  double lane_length = o->net.lane_len_pytsc[l] / o->cfg.veh_size_min_gap;
  return r->lane_n[l] / lane_length;
}
/* value of bin `b` of the visibility-padded position matrix (retriever.py:45-51) */
static inline double posmat_at(const orc_t *o, const rep_t *r, int l, int b) {
  int bins = o->net.lane_bins[l];
  if (r->lane_n[l] == 0 || bins <= 0) return -1.0; /* [-1]*visibility */
  if (b >= bins) return -1.0;                      /* padding */
  return r->lane_posmat[(size_t)l * o->max_bins + b];
}
static inline int posmat_len(const orc_t *o, const rep_t *r, int l) {
  int bins = o->net.lane_bins[l], vis = o->cfg.visibility;
  if (r->lane_n[l] == 0 || bins <= 0) return vis;
  return bins < vis ? vis : bins;
}

/* TLSFree/RoundRobinPhaseSelectLogic (common/traffic_signal.py:329-361,375-404) */
static void allowable_switches(const orc_t *o, const rep_t *r, int a, int *mask) {
  const tsc_net_t *n = &o->net;
  int np_ = n->agent_n_phases[a];
  const int *green = n->agent_phase_green + a * TSC_MAX_AGENT_PHASES;
  int cur = r->cur_idx[a], t = r->time_on_phase[a];
  int nxt = (cur + 1) % np_;
  for (int i = 0; i < np_; i++) mask[i] = 0;
  if (green[cur]) {
    int mn = n->agent_phase_min[a * TSC_MAX_AGENT_PHASES + cur];
    int mx = n->agent_phase_max[a * TSC_MAX_AGENT_PHASES + cur];
    if (t < mn) {
      mask[cur] = 1;
    } else if (t >= mn && t < mx) {
      mask[cur] = 1;
      mask[nxt] = 1;
    } else if (t == mx) {
      mask[nxt] = 1;
    } /* t > max: the reference hits breakpoint(); all-zero mask here */
  } else {
    if (o->cfg.round_robin) {
      mask[nxt] = 1;
    } else {
      for (int i = 0; i < np_; i++)
        if (green[i] && i != cur - 1) mask[i] = 1;
    }
  }
}

static int n_actions(const orc_t *o) {
  if (o->cfg.action_space == TSC_ACT_PHASE_SWITCH) return 2;
  int m = 0;
  for (int a = 0; a < o->net.n_agents; a++)
    if (o->net.agent_n_phases[a] > m) m = o->net.agent_n_phases[a];
  return m;
}

static void fill_outputs(orc_t *o, int ri, const tsc_step_out_t *out) {
  const tsc_net_t *n = &o->net;
  rep_t *r = &o->rep[ri];
  int A = n->n_agents, vis = o->cfg.visibility;
  int obs_size = TSC_MAX_CTRL_LANES * (vis + TSC_LANE_STATIC) + TSC_MAX_PHASES;
  int state_size = TSC_MAX_CTRL_LANES * (3 + TSC_LANE_STATIC) + TSC_MAX_PHASES;
  int nact = n_actions(o);
  measure_lanes(o, r);
  double *local = (double *)malloc(sizeof(double) * (size_t)A * 2);
  double *pressures = local + A;
  int nq_agent_sum_dummy = 0;
  (void)nq_agent_sum_dummy;
  int sim_step = r->step - o->cfg.initial_wait_time;
  int n_changed = 0;
  for (int a = 0; a < A; a++) {
    /* TrafficSignal.update_stats (backends/cityflow/traffic_signal.py:101-141) */
    int ib = n->agent_inc_ptr[a], ie = n->agent_inc_ptr[a + 1];
    int ob = n->agent_out_ptr[a], oe = n->agent_out_ptr[a + 1];
    int nq = 0;
    double occ = 0, ms = 0, md = 0;
    for (int k = ib; k < ie; k++) {
      int l = n->agent_inc_lanes[k];
      nq += r->lane_q[l];
      occ += lane_occupancy(o, r, l);
      ms += lane_mean_speed(r, l);
      md += 1 - lane_mean_speed(r, l) / n->drv_max_speed[l];
    }
    occ /= (ie - ib);
    ms /= (ie - ib);
    md /= (ie - ib);
    double oocc = 0;
    for (int k = ob; k < oe; k++) oocc += lane_occupancy(o, r, n->agent_out_lanes[k]);
    oocc /= (oe - ob);
    double pressure = fabs(occ - oocc);
    pressures[a] = pressure;
    int cur = r->cur_idx[a];
    int raw = n->agent_phase_raw[a * TSC_MAX_AGENT_PHASES + cur];
    (void)raw;
    double norm_top = (double)r->time_on_phase[a] / (double)n->agent_phase_max[a * TSC_MAX_AGENT_PHASES + cur];
    if (r->time_on_phase[a] == 0) norm_top = 0; /* set_initial_phase */
    n_changed += r->phase_changed[a] ? 1 : 0;
    /* local reward before neighbour mixing (common/reward.py:76-80 / :124-128) */
    double fc = o->cfg.flickering_coef;
    double pc = r->phase_changed[a] ? 1.0 : 0.0;
    if (o->cfg.reward_kind == 0) local[a] = -fc * pc - nq - 1e-6;
    else local[a] = -fc * pc - pressure - 1e-6;
    if (out->agent_stats_dev) {
      double *s = out->agent_stats_dev + ((size_t)ri * A + a) * TSC_N_AGENT_STATS;
      s[TSC_A_N_QUEUED] = nq;
      s[TSC_A_OCCUPANCY] = occ;
      s[TSC_A_MEAN_SPEED] = ms;
      s[TSC_A_MEAN_DELAY] = md;
      s[TSC_A_OUT_OCCUPANCY] = oocc;
      s[TSC_A_PRESSURE] = pressure;
      s[TSC_A_NORM_TIME_ON_PHASE] = norm_top;
      s[TSC_A_SIM_STEP_NORM] = sim_step / 3600.0;
    }
    if (out->phase_dev) {
      int32_t *p = out->phase_dev + ((size_t)ri * A + a) * 3;
      p[0] = cur;
      p[1] = r->time_on_phase[a];
      p[2] = r->phase_changed[a];
    }
    /* What the reference's rule-based controllers compute per phase from the position matrices
     * (pytsc/controllers/controllers.py): GreedyController._compute_queue_for_phase :93-112 -- bins == 0.0 of
     * pos_mat[-visibility:] over the phase's incoming lanes; SOTLController._compute_flow_for_phase :227-242 -- bins >= 0.0
     * of the same windows; MaxPressureController._compute_pressure_for_phase :153-178 -- sum over the incoming lanes of
     * |vehicles in its window - vehicles in pos_mat[:visibility] of the LAST outgoing lane listed for it| (the loop over
     * out_lanes overwrites, :171-176).  The movements come from tsc_net_t::agent_phase_moves. */
    if (out->phase_scores_dev && n->agent_phase_moves) {
      int np_s = n->agent_n_phases[a];
      for (int p_ = 0; p_ < TSC_MAX_PHASES; p_++) {
        int sq = 0, sf = 0, sp = 0;
        if (p_ < np_s) {
          const uint32_t *mv = n->agent_phase_moves + ((size_t)a * TSC_MAX_AGENT_PHASES + p_) * 4;
          unsigned long long sel = (unsigned long long)mv[1] | ((unsigned long long)mv[2] << 32);
          for (int k = 0; k < ie - ib && k < 16; k++) {
            if (!((mv[0] >> k) & 1u)) continue;
            int l = n->agent_inc_lanes[ib + k];
            int lo = n->agent_out_lanes[ob + (int)((sel >> (4 * k)) & 15ull)];
            int plen = posmat_len(o, r, l), n_in = 0, n_z = 0, n_o = 0;
            for (int b = plen - vis; b < plen; b++) {
              double v = posmat_at(o, r, l, b);
              n_z += v == 0.0;
              n_in += v >= 0.0;
            }
            for (int b = 0; b < vis; b++) n_o += posmat_at(o, r, lo, b) >= 0.0;
            sq += n_z;
            sf += n_in;
            sp += abs(n_in - n_o);
          }
        }
        int16_t *ps = out->phase_scores_dev + (((size_t)ri * A + a) * TSC_MAX_PHASES + p_) * 3;
        ps[0] = (int16_t)sq;
        ps[1] = (int16_t)sf;
        ps[2] = (int16_t)sp;
      }
    }
    /* PositionMatrix.get_observations (common/observations.py:140-160) with
     * the pad_list integer truncation (utils.py:91-112) and the >0 compaction
     * (observations.py:72-88) */
    if (out->obs_dev) {
      float *ob_ = out->obs_dev + ((size_t)ri * A + a) * obs_size;
      int w = 0;
      for (int k = ib; k < ie; k++) {
        int l = n->agent_inc_lanes[k];
        for (int j = 0; j < TSC_LANE_STATIC; j++) ob_[w++] = (float)n->lane_static[l * TSC_LANE_STATIC + j];
        int plen = posmat_len(o, r, l);
        for (int b = plen - vis; b < plen; b++) { /* pos_mat[-visibility:] */
          if (b < 0) continue;
          double val = posmat_at(o, r, l, b);
          if (val > 0) {
            double c = val; /* + N(0, 0) */
            if (c < 0.0) c = 0.0;
            if (c > 1.0) c = 1.0;
            ob_[w++] = (float)(long long)c; /* truncation toward zero */
          }
        }
      }
      for (; w < obs_size - TSC_MAX_PHASES; w++) ob_[w] = -1.0f;
      int np_ = n->agent_n_phases[a];
      for (int i = 0; i < TSC_MAX_PHASES; i++) ob_[w++] = i < np_ ? (i == cur ? 1.0f : 0.0f) : -1.0f;
    }
    /* get_state (common/observations.py:183-205) */
    if (out->state_dev) {
      float *st = out->state_dev + ((size_t)ri * A + a) * state_size;
      int w = 0;
      for (int k = ib; k < ie; k++) {
        int l = n->agent_inc_lanes[k];
        for (int j = 0; j < TSC_LANE_STATIC; j++) st[w++] = (float)n->lane_static[l * TSC_LANE_STATIC + j];
        st[w++] = (float)r->lane_q[l];
        st[w++] = (float)(long long)lane_occupancy(o, r, l);
        st[w++] = (float)(long long)lane_mean_speed(r, l);
      }
      for (; w < state_size - TSC_MAX_PHASES; w++) st[w] = -1.0f;
      int np_ = n->agent_n_phases[a];
      for (int i = 0; i < TSC_MAX_PHASES; i++) st[w++] = (i < np_ && i == cur) ? 1.0f : 0.0f;
    }
    /* get_mask (common/actions.py:119-131,169-188) */
    if (out->mask_dev) {
      int m[TSC_MAX_AGENT_PHASES];
      allowable_switches(o, r, a, m);
      uint8_t *mk = out->mask_dev + ((size_t)ri * A + a) * nact;
      int np_ = n->agent_n_phases[a];
      if (o->cfg.action_space == TSC_ACT_PHASE_SWITCH) {
        mk[0] = (uint8_t)(m[cur] ? 1 : 0);
        mk[1] = (uint8_t)(m[(cur + 1) % np_] ? 1 : 0);
      } else {
        for (int i = 0; i < nact; i++) mk[i] = (uint8_t)(i < np_ ? m[i] : 0);
      }
    }
  }
  /* local rewards with neighbour mixing (common/reward.py:81-88) */
  for (int a = 0; a < A; a++) {
    double rew = local[a];
    for (int k = n->agent_nbr_ptr[a]; k < n->agent_nbr_ptr[a + 1]; k++)
      rew += n->agent_nbr_w[k] * local[n->agent_nbr[k]];
    if (out->local_reward_dev) out->local_reward_dev[(size_t)ri * A + a] = (float)rew;
    if (out->local_reward64_dev) out->local_reward64_dev[(size_t)ri * A + a] = rew;
  }
  /* network metrics (backends/cityflow/metrics.py:24-260) */
  long long nq_all = 0, nveh_all = 0;
  double dens = 0, nms = 0, tvs = 0;
  for (int l = 0; l < n->n_lanes; l++) {
    nq_all += r->lane_q[l];
    nveh_all += r->lane_n[l];
    dens += lane_occupancy(o, r, l);
    nms += lane_mean_speed(r, l) / n->drv_max_speed[l];
    tvs += lane_mean_speed(r, l) * r->lane_n[l];
  }
  double density = dens / n->n_lanes;
  double norm_mean_speed = nms / n->n_lanes;
  double mean_speed = nveh_all == 0 ? 0.0 : tvs / (double)nveh_all;
  double pressure = np_pairwise_sum(pressures, A);
  double flick = (double)n_changed / (double)A;
  double g;
  if (o->cfg.reward_kind == 0) {
    g = 1e-6;
    g += o->cfg.flickering_coef * flick;
    g += (double)nq_all;
    g = -1 * g;
  } else {
    g = 1e-6;
    g -= o->cfg.flickering_coef * flick;
    g -= pressure;
  }
  if (out->global_reward_dev) out->global_reward_dev[ri] = (float)g;
  if (out->global_reward64_dev) out->global_reward64_dev[ri] = g;
  if (out->stats_dev) {
    double *s = out->stats_dev + (size_t)ri * TSC_N_STATS;
    s[TSC_S_TIME_STEP] = r->step * o->cfg.interval;
    s[TSC_S_AVG_TRAVEL_TIME] = tsco_engine_average_travel_time(o, ri);
    s[TSC_S_N_QUEUED] = (double)nq_all;
    s[TSC_S_MEAN_SPEED] = mean_speed;
    s[TSC_S_MEAN_DELAY] = 1 - norm_mean_speed;
    s[TSC_S_DENSITY] = density;
    s[TSC_S_PRESSURE] = pressure;
    s[TSC_S_NETWORK_FLOW] = density * norm_mean_speed;
  }
  /* episode_over / is_terminated (pytsc/__init__.py:58-62; simulator.py:20-30) */
  if (out->done_dev) out->done_dev[ri] = (uint8_t)(sim_step > 0 && sim_step % o->cfg.episode_limit == 0);
  if (out->terminated_dev) out->terminated_dev[ri] = (uint8_t)(sim_step == o->cfg.sim_length);
  free(local);
}

/* ---------------------------------------------------- env-level API (host) */
int tsco_reset_sets(void *h, const uint8_t *mask, const int32_t *sets) {
  orc_t *o = (orc_t *)h;
  for (int i = 0; i < o->cfg.n_replicas; i++) {
    if (mask && !mask[i]) continue;
    rep_reset(o, &o->rep[i], i, 1, sets ? sets[i] : -1);
    for (int s = 0; s < o->cfg.initial_wait_time; s++) engine_step(o, &o->rep[i]);
  }
  return 0;
}

int tsco_reset(void *h, const uint8_t *mask) { return tsco_reset_sets(h, mask, NULL); }

int tsco_observe(void *h, const tsc_step_out_t *out) {
  orc_t *o = (orc_t *)h;
  for (int i = 0; i < o->cfg.n_replicas; i++) fill_outputs(o, i, out);
  return 0;
}

/* action_space.apply (common/actions.py:99-108,144-158) -> switch_phase
 * (backends/cityflow/traffic_signal.py:51-59) -> update_current_phase
 * (common/traffic_signal.py:94-109); then simulator_step
 * (backends/cityflow/simulator.py:80-89) */
int tsco_step(void *h, const int32_t *actions, const tsc_step_out_t *out) {
  orc_t *o = (orc_t *)h;
  const tsc_net_t *n = &o->net;
  for (int i = 0; i < o->cfg.n_replicas; i++) {
    rep_t *r = &o->rep[i];
    for (int a = 0; a < n->n_agents; a++) {
      int act = actions ? actions[(size_t)i * n->n_agents + a] : -1;
      if (act == -1) continue; /* signal left untouched */
      int np_ = n->agent_n_phases[a];
      int target;
      if (o->cfg.action_space == TSC_ACT_PHASE_SWITCH) target = act == 1 ? (r->cur_idx[a] + 1) % np_ : r->cur_idx[a];
      else target = act;
      if (target < 0 || target >= np_) return TSC_EINVAL;
      if (target == r->cur_idx[a]) {
        r->phase_changed[a] = 0;
        r->time_on_phase[a] += o->cfg.yellow_time;
      } else {
        r->phase_changed[a] = 1;
        r->time_on_phase[a] = o->cfg.yellow_time;
      }
      r->cur_idx[a] = target;
      r->raw_phase[a] = n->agent_phase_raw[a * TSC_MAX_AGENT_PHASES + target];
    }
    r->have_phase_changed = 1;
    for (int s = 0; s < o->cfg.delta_time; s++) engine_step(o, r);
    if (out) fill_outputs(o, i, out);
  }
  return 0;
}

int tsco_lane_detail(void *h, int32_t *lane_counts, double *lane_speed, double *posmat, int32_t max_bins) {
  orc_t *o = (orc_t *)h;
  int nl = o->net.n_lanes;
  for (int i = 0; i < o->cfg.n_replicas; i++) {
    rep_t *r = &o->rep[i];
    for (int l = 0; l < nl; l++) {
      lane_counts[((size_t)i * nl + l) * 2 + 0] = r->lane_n[l];
      lane_counts[((size_t)i * nl + l) * 2 + 1] = r->lane_q[l];
      lane_speed[(size_t)i * nl + l] = r->lane_sum_speed[l];
      if (posmat)
        for (int b = 0; b < max_bins; b++)
          posmat[((size_t)i * nl + l) * max_bins + b] = posmat_at(o, r, l, b);
    }
  }
  return 0;
}

/* controllers' window counts (controllers/controllers.py:93-242) from the full matrices */
int tsco_lane_window_counts(void *h, int32_t *out) {
  orc_t *o = (orc_t *)h;
  int nl = o->net.n_lanes, vis = o->cfg.visibility;
  for (int i = 0; i < o->cfg.n_replicas; i++) {
    rep_t *r = &o->rep[i];
    for (int l = 0; l < nl; l++) {
      int plen = posmat_len(o, r, l), nz = 0, no = 0, nout = 0;
      for (int b = plen - vis; b < plen; b++) {
        double v = posmat_at(o, r, l, b);
        nz += v == 0.0;
        no += v >= 0.0;
      }
      for (int b = 0; b < vis; b++) nout += posmat_at(o, r, l, b) >= 0.0;
      int32_t *p = out + ((size_t)i * nl + l) * 3;
      p[0] = nz;
      p[1] = no;
      p[2] = nout;
    }
  }
  return 0;
}

int tsco_snapshot(void *h, int32_t rep, int32_t cap, int32_t *drv, double *dist, double *speed, int32_t *vid) {
  orc_t *o = (orc_t *)h;
  rep_t *r = &o->rep[rep];
  int c = 0;
  for (int d = 0; d < o->n_drv; d++)
    for (int i = 0; i < r->drv[d].n; i++) {
      if (c < cap) {
        const veh_t *v = &r->veh[r->drv[d].v[i]];
        drv[c] = d;
        dist[c] = v->dist;
        speed[c] = v->speed;
        vid[c] = r->drv[d].v[i];
      }
      c++;
    }
  return c;
}

int tsco_n_actions(void *h) { return n_actions((orc_t *)h); }
int tsco_max_bins(void *h) { return ((orc_t *)h)->max_bins; }
