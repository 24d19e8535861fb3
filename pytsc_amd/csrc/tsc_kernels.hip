// Hand-written HIP kernels (gfx950 / CDNA4) of the batched traffic microsimulator:
// the hot path TrafficSignalNetwork.step() (reference pytsc/__init__.py:178-182).
//
// Decomposition: kernels that walk state in order put 64 consecutive
// replicas of ONE drivable (lane or lane-link) or ONE agent on the lanes of a wavefront,
// so HBM accesses coalesce along the replica axis and topology is wave-uniform (scalar
// loads, no divergence on the network structure); the arithmetic runs on dense batches
// of 64 work items drawn from all replicas.  One engine sub-step is five launches
// (Jacobi update: decisions read the committed state):
//   k_notify  (agent    x replica)  cross occupancy masks per intersection
//   k_enum    (drivable x replica)  classification: settled vehicles are finished here,
//                                   the others go to work lists spanning all replicas
//   k_plan    (list)                list counts -> batch table
//   k_exec    (batch of 64 items)   leader/gap, car-following, signal gating, cross
//                                   yielding -> buffered speed/position, outbox
//   k_commit  (drivable x replica)  compaction, hand-off pull from feeder drivables,
//                                   retire, spawn for the next step
// and one env step ends with k_measure (lane x replica), k_agent (agent x
// replica, observations staged through LDS for coalesced [R, A, obs] stores)
// and k_stats (replica).  No MFMA: this is integer / f64 scalar work.
//
// f64 arithmetic follows the same operation order as the CPU oracle
// (compiled with -ffp-contract=off on both sides) so results are bit-identical.
#include "tsc_device.h"

// Diagnostic ablation switches (tools/ablate_action.py) exist only in -DTSC_DIAG builds; the
// product library has none of them in its kernels.
#ifdef TSC_DIAG
#define TSC_ABL(bits) (C.ablate & (bits))
#define TSC_ABLP(v, bits) ((v) & (bits))
#define TSC_ABLV C.ablate
#else
#define TSC_ABL(bits) 0
#define TSC_ABLP(v, bits) 0
#define TSC_ABLV 0
#endif

// 32-bit index arithmetic: tsc_create refuses tables with more than 2^31 - 1 elements per array
#define IDX(row) ((unsigned int)(row) * (unsigned int)S.RS + (unsigned int)r)
// Vehicle-slot arrays are tiled by 64-replica block: [replica block][slot row][64 lanes].
// One wavefront's slots of one lane are then a few KB of contiguous memory per field
// (consecutive rows 512 B apart instead of RS*8 bytes): DRAM-page and TLB friendly.
#define SIDX(row) ((((unsigned int)r >> 6) * (unsigned int)S.n_slots + (unsigned int)(row)) * 64u + ((unsigned int)r & 63u))
// {dist, speed} records: TSC_ROWPACK consecutive rows of a replica sit side by side, so a vehicle
// and its leader -- and the neighbours of a platoon -- share 128-byte lines more often than
// replicas do (rows are only about one third occupied by moving vehicles).  Slot bases are
// multiples of TSC_ROWPACK (tsc_create).
#define DIDX(row)                                                                                        \
  (((((unsigned int)r >> 6) * ((unsigned int)S.n_slots / TSC_ROWPACK) + ((unsigned int)(row) / TSC_ROWPACK)) * 64u + \
    ((unsigned int)r & 63u)) * TSC_ROWPACK + ((unsigned int)(row) % TSC_ROWPACK))

struct Tm {
  double len, maxPosAcc, maxNegAcc, usualNegAcc, minGap, maxSpeed, headway, yieldDist, turnSpeed, approach;
  // derived per template at create (same f64 operations, host side): see TSC_DT_* in tsc_device.h
  double hA_max, iA_max, hA_usual, iA_usual, limit, hwden;
  // Divisors known at compile time (the vehicle-specialised instantiations of kr_step): their correctly rounded
  // reciprocals, and whether div_c may use them (false everywhere else: the test folds away)
  bool cd = false;
  double r_maxNegAcc = 0, r_usualNegAcc = 0, r_maxSpeed = 0, r_turnSpeed = 0;
};

// a / b, correctly rounded, for a divisor whose correctly rounded reciprocal rb = RN(1 / b) is at hand (a compile-time
// constant in the callers): q0 = RN(a rb) is within 2 ulp of a / b; one Newton step on the exact residual (fma) makes it
// faithful; Markstein's theorem (IBM J. R&D 34, 1990: y = RN(1 / b), q faithful, r = a - b q exact => RN(q + r y) = RN(a / b),
// for every b whose significand is not all ones) makes the second step's result THE correctly rounded quotient -- the
// number the hardware's v_div_scale / v_rcp / v_fma ... v_div_fixup sequence delivers, in five dependent operations
// instead of about twelve.  Valid away from overflow / underflow of the residuals (|a| in [1e-290, 1e290] or a = +0: the
// callers divide squared speeds and distances) -- tests/test_divconst.py checks it against the division on the host.
__device__ __forceinline__ double div_c(double a, double b, double rb) {
  double q = a * rb;
  double r = fma(-b, q, a);
  q = fma(r, rb, q);
  r = fma(-b, q, a);
  return fma(r, rb, q);
}

// a / b through div_c when the host found every lane's divisors fit for it (RC: a template switch of the caller, chosen by
// DevNet::recip_ok), else the division: the same bits either way.  a: a count or a mean speed (0 or within [1e-290, 1e290]).
template <bool RC>
__device__ __forceinline__ double div_r(double a, double b, double rb) {
  return RC ? div_c(a, b, rb) : a / b;
}

__device__ __forceinline__ Tm load_tm_p(const double *tbl, int id);
__device__ __forceinline__ Tm load_tm(const DevNet &N, int id) { return load_tm_p((const double *)N.tmpl, id); }
__device__ __forceinline__ Tm load_tm_p(const double *tbl, int id) {
  const double *p = tbl + (size_t)id * TSC_DT_STRIDE;
  Tm t;
  t.len = p[TSC_T_LEN];
  t.maxPosAcc = p[TSC_T_MAXPOSACC];
  t.maxNegAcc = p[TSC_T_MAXNEGACC];
  t.usualNegAcc = p[TSC_T_USUALNEGACC];
  t.minGap = p[TSC_T_MINGAP];
  t.maxSpeed = p[TSC_T_MAXSPEED];
  t.headway = p[TSC_T_HEADWAY];
  t.yieldDist = p[TSC_T_YIELDDIST];
  t.turnSpeed = p[TSC_T_TURNSPEED];
  t.approach = p[TSC_T_APPROACH];
  t.hA_max = p[TSC_DT_HA_MAX];
  t.iA_max = p[TSC_DT_IA_MAX];
  t.hA_usual = p[TSC_DT_HA_USUAL];
  t.iA_usual = p[TSC_DT_IA_USUAL];
  t.limit = p[TSC_DT_LIMIT];
  t.hwden = p[TSC_DT_HWDEN];
  return t;
}

template <bool SINGLE>
__device__ __forceinline__ int tmpl_of(const DevNet &N, int vbase, int vid) {
  return SINGLE ? 0 : N.veh_tmpl[vbase + vid];
}

__device__ __forceinline__ double min2(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double max2(double a, double b) { return a > b ? a : b; }

// C++ (int)x as x86-64 does it: out of range -> INT_MIN
__device__ __forceinline__ int cvt_int(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
  return (int)x;
}

// ---------------------------------------------------------------- kinematics
__device__ __forceinline__ double min_brake_distance(double speed, const Tm &t) {
  if (t.cd) return div_c(0.5 * speed * speed, t.maxNegAcc, t.r_maxNegAcc);
  return 0.5 * speed * speed / t.maxNegAcc;
}
__device__ __forceinline__ bool can_yield(double speed, const Tm &t, double dist) {
  return (dist > 0 && min_brake_distance(speed, t) < dist - t.yieldDist) || (dist < 0 && dist + t.len < 0);
}
__device__ __forceinline__ double distance_until_speed(double cur, double speed, double acc, double interval) {
  if (speed <= cur) return 0;
  int stage1steps = (int)floor((speed - cur) / acc / interval);
  double stage1speed = cur + stage1steps * acc / interval;
  double stage1dis = (cur + stage1speed) * (stage1steps * interval) / 2;
  return stage1dis + (stage1speed < speed ? ((stage1speed + speed) * interval / 2) : 0);
}
__device__ __forceinline__ int reach_steps(double cur, double distance, double target, double acc, double interval) {
  if (distance <= 0) return 0;
  if (cur > target) return (int)ceil(distance / cur);
  double dut = distance_until_speed(cur, target, acc, interval);
  if (dut > distance) {
    return (int)ceil((sqrt(cur * cur + 2 * acc * distance) - cur) / acc / interval);
  } else {
    return (int)ceil((target - cur) / acc / interval) + (int)ceil((distance - dut) / target / interval);
  }
}
// a = 0.5 / dF and ia = 0.5 / a of the follower's deceleration dF come from the template table; lb = 0.5 vL^2 / dL, the
// leader's braking distance, from the caller
__device__ __forceinline__ double no_collision_speed(double vL, double dL, double lb, double vF, double a, double ia, double gap,
                                                     double interval, double targetGap) {
  double c = vF * interval / 2 + targetGap - lb - gap;
  double b = 0.5 * interval;
  if (b * b < 4 * a * c) return -100;
  double v1 = ia * (sqrt(b * b - 4 * a * c) - b);
  double v2 = 2 * vL - dL * interval + 2 * (gap - targetGap) / interval;
  return min2(v1, v2);
}
__device__ __forceinline__ double car_follow_speed(double speed, const Tm &t, double vL, const Tm &tl, double gap,
                                                   double interval) {
  // the leader's braking distance at its maximal and at its usual deceleration: the same division when the two are equal
  // (CityFlow's default vehicle: 4.5 both) -- with one template the test is wave-uniform and the second division is skipped
  const double hv = 0.5 * vL * vL;
  const double lb_max = tl.cd ? div_c(hv, tl.maxNegAcc, tl.r_maxNegAcc) : hv / tl.maxNegAcc;
  const double lb_usual = tl.usualNegAcc == tl.maxNegAcc ? lb_max : hv / tl.usualNegAcc;
  // The bound at the maximal deceleration (target gap 0) is never below the one at the usual deceleration (target gap
  // minGap >= 0) when the two decelerations are equal, on the leader's side and on the follower's: the two evaluations then
  // differ in the target gap alone, and every operation of no_collision_speed is monotone in it, rounding included -- so
  // min(first, second) IS the second (tests/test_measure_fastpath.py checks it on the host).  With one vehicle template the
  // test is wave-uniform and the first evaluation -- a square root and a division -- is skipped.
  const bool same = tl.usualNegAcc == tl.maxNegAcc && t.usualNegAcc == t.maxNegAcc && t.minGap >= 0.0;
  double assumeDecel = 0, leaderSpeed = vL;
  if (speed > leaderSpeed) assumeDecel = speed - leaderSpeed;
  double s = no_collision_speed(vL, tl.usualNegAcc, lb_usual, speed, t.hA_usual, t.iA_usual, gap, interval, t.minGap);
  if (!same) {
    const double s_max = no_collision_speed(vL, tl.maxNegAcc, lb_max, speed, t.hA_max, t.iA_max, gap, interval, 0);
    s = min2(s_max, s);
  }
  s = min2(s, (gap + (leaderSpeed + assumeDecel / 2) * interval - speed * interval / 2) / t.hwden);
  return s;
}
// distance covered by one more acceleration step followed by usual braking to a stop
__device__ __forceinline__ double brake_distance_after_accel(double speed, const Tm &t, double interval) {
  double nextSpeed = speed + t.maxPosAcc * interval;
  if (t.cd) return (speed + nextSpeed) * interval / 2 + (div_c(nextSpeed * nextSpeed, t.usualNegAcc, t.r_usualNegAcc) / 2);
  return (speed + nextSpeed) * interval / 2 + (nextSpeed * nextSpeed / t.usualNegAcc / 2);
}
__device__ __forceinline__ double stop_before_speed(double speed, const Tm &t, double distance, double interval) {
  double bd = brake_distance_after_accel(speed, t, interval);
  if (bd < distance) return speed + t.maxPosAcc * interval;
  double takeInterval = 2 * distance / (speed + 1e-9) / interval;
  // (one division for both cases: the divisor is (int)takeInterval as a double, or takeInterval itself)
  const double den = takeInterval >= 1 ? (double)cvt_int(takeInterval) : takeInterval;
  return speed - speed / den;
}

// XCD-aware block -> (replica block, row block) map: blocks b and b+8 share an
// XCD (MI355X_MICROARCH.md, Workgroup dispatch), so all row-blocks of one
// 64-replica block go to the same XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ bool map_block(int n_rb, int n_rowblk, int &rb, int &rowblk) {
  int bid = blockIdx.x;
  int xcd = bid & 7;
  int j = bid >> 3;
  rb = (j / n_rowblk) * 8 + xcd;
  rowblk = j % n_rowblk;
  return rb < n_rb;
}
static inline int grid_for(int n_rb, int n_rowblk) { return ((n_rb + 7) / 8) * 8 * n_rowblk; }

// ------------------------------------------------------------------ k_notify
// Per intersection: which lane-links currently have (E) a vehicle that just
// left them as the tail of their end lane, (V) vehicles on them, (S) the head of
// their start lane heading for them.  Restates the candidate set of the engine's
// notifyCross as three 64-bit masks; the exact
// per-cross foe is resolved lazily by find_foe() only for occupied links.
#define NOTIFY_WAVES 12
#define NOTIFY_LANES 24 /* incoming / outgoing lanes of an intersection staged in LDS (more: direct loads) */
// One workgroup = one intersection x 64 replicas: its waves split the intersection's
// lane-links, each lane-link's three candidate bits are OR-ed in LDS, wave 0 writes
// the masks (coalesced).
// Round 4: this kernel REBUILDS the masks -- after a reset (list_reset) -- and no longer runs once per sub-step: between two
// sub-steps the masks change only where k_commit changed a drivable, and k_commit updates the bits there (see its end).  It
// was a tenth of a 16x16 env step (0.25 ms x 5) for 0.2 GB of traffic.  `tick`: the per-sub-step duties it used to have
// beside the masks (the step index, the work-list counters; diagnostic launches only: k_enum and a memset have them now).
__global__ __launch_bounds__(64 * NOTIFY_WAVES) void k_notify(DevNet N, DevState S, int n_rb, int n_agents, int tick) {
  __shared__ unsigned long long m_e[64], m_v[64], m_s[64];
  __shared__ int s_a[2 * NOTIFY_LANES][64], s_b[2 * NOTIFY_LANES][64]; // lane words: count, {aux of the head | previous drivable of the tail}
  int rb, a;
  const bool live = map_block(n_rb, n_agents, rb, a);
  a = __builtin_amdgcn_readfirstlane(a);
  const int lx = threadIdx.x, wy = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
  if (tick && blockIdx.x == 0) // the work-list counters of the sub-step that starts here
    for (int i = threadIdx.y * 64 + threadIdx.x; i < S.n_lists + 2; i += 64 * NOTIFY_WAVES) S.wl_nb[i] = 0;
  const int r = rb * 64 + lx;
  const bool mine = live && r < S.R;
  if (wy == 0) {
    m_e[lx] = 0;
    m_v[lx] = 0;
    m_s[lx] = 0;
    if (tick && mine && a == 0) S.t[r] += 1; // the step index being executed from here on
  }
  // the words of the intersection's lanes are read ONCE per workgroup into LDS (three lane-links share a start lane, three
  // an end lane: reading them per lane-link was 180 rows of 256 bytes per workgroup, this is 84) -- this kernel is bound by
  // exactly these reads
  const int ib = N.agent_inc_ptr[a], n_inc = N.agent_inc_ptr[a + 1] - ib;
  const int ob = N.agent_out_ptr[a], n_out = N.agent_out_ptr[a + 1] - ob;
  const bool staged = n_inc <= NOTIFY_LANES && n_out <= NOTIFY_LANES; // (uniform; else: the direct loads below)
  // Round 4: every word this workgroup reads is requested before the first one is used -- the lane words of up to four
  // staging rounds and the counts of this wavefront's lane-links (at most 64 per intersection: six rounds of twelve
  // wavefronts) -- one round trip to memory per workgroup instead of three (a round of lane words, the next one behind its
  // LDS stores, then the links' counts behind the barrier).
  constexpr int NR = (2 * NOTIFY_LANES + NOTIFY_WAVES - 1) / NOTIFY_WAVES, NK = (64 + NOTIFY_WAVES - 1) / NOTIFY_WAVES;
  const int lb = N.agent_link_ptr[a], le = N.agent_link_ptr[a + 1];
  int va[NR], vb[NR], cl_r[NK];
  if (mine && staged) {
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int j = wy + i * NOTIFY_WAVES;
      va[i] = vb[i] = 0;
      if (j < n_inc) {
        const int sl = N.agent_inc_lanes[ib + j];
        va[i] = S.cnt[IDX(sl)];
        vb[i] = S.ab[SIDX(N.drv_slot_base[sl])].x;
      } else if (j < n_inc + n_out) {
        const int e = N.agent_out_lanes[ob + j - n_inc];
        va[i] = S.cnt[IDX(e)];
        vb[i] = S.tail_prev[IDX(e)];
      }
    }
  }
  if (mine) {
#pragma unroll
    for (int i = 0; i < NK; i++) {
      const int li = lb + wy + i * NOTIFY_WAVES;
      cl_r[i] = li < le ? S.cnt[IDX(N.agent_links[li])] : 0;
    }
  }
  if (mine && staged) {
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int j = wy + i * NOTIFY_WAVES;
      if (j < n_inc) {
        s_a[j][lx] = va[i];
        s_b[j][lx] = vb[i];
      } else if (j < n_inc + n_out) {
        s_a[NOTIFY_LANES + j - n_inc][lx] = va[i];
        s_b[NOTIFY_LANES + j - n_inc][lx] = vb[i];
      }
    }
  }
  __syncthreads();
  if (mine) {
    unsigned long long oe = 0, ov = 0, os = 0;
#pragma unroll
    for (int i = 0; i < NK; i++) {
      const int li = lb + wy + i * NOTIFY_WAVES;
      if (li >= le) break;
      const int link = N.agent_links[li];
      const int k = link - N.n_lanes;
      const unsigned long long bit = 1ull << (li - lb);
      const int cl = cl_r[i];
      int he, tpv, hs, ax;
      if (staged) {
        const int2 lo = N.link_lane_local[k]; // positions of its start / end lane among the intersection's lanes
        hs = s_a[lo.x][lx];
        ax = s_b[lo.x][lx];
        he = s_a[NOTIFY_LANES + lo.y][lx];
        tpv = s_b[NOTIFY_LANES + lo.y][lx];
      } else {
        const int e = N.link_end_lane[k];
        const int sl = N.link_start_lane[k];
        he = S.cnt[IDX(e)];
        tpv = S.tail_prev[IDX(e)];
        hs = S.cnt[IDX(sl)];
        ax = S.ab[SIDX(N.drv_slot_base[sl])].x;
      }
      if ((cl & TSC_CNT_MASK) > 0) ov |= bit;
      if ((he & TSC_CNT_MASK) > 0 && !(he & TSC_NEW_BIT) && tpv == link) oe |= bit;
      if ((hs & TSC_CNT_MASK) > 0 && ax == link) os |= bit;
    }
    if (oe) atomicOr(&m_e[lx], oe);
    if (ov) atomicOr(&m_v[lx], ov);
    if (os) atomicOr(&m_s[lx], os);
  }
  __syncthreads();
  if (mine && wy == 0) {
    AgentRec *ar = &S.arec[IDX(a)];
    ar->oe = m_e[lx];
    ar->ov = m_v[lx];
    ar->os = m_s[lx];
  }
}

// ------------------------------------------------------------------ find_foe
struct Foe {
  double d2, speed;
  int vid, ell, tmpl;
};

// The vehicle the engine's notifyCross registers for the
// cross at distance `xo` on lane-link `other` (wave-uniform): the first of
//   E  the tail of the end lane, if it came from this link,
//   V  the vehicles on the link, front to back,
//   S  the head of the start lane, if it is heading here on green,
// whose rear has not yet cleared the cross.  Resolved lazily, only for links
// whose bit is set in the per-intersection occupancy masks.
template <bool SINGLE>
__device__ __forceinline__ Foe find_foe(const DevNet &N, const DevState &S, int r, int vbase, int other, int bitpos,
                                        double xo, unsigned long long oe, unsigned long long ov,
                                        unsigned long long os_green) {
  Foe f;
  f.d2 = 0;
  f.speed = 0;
  f.vid = -1;
  f.ell = 0x7fffffff;
  f.tmpl = 0;
  const int k = other - N.n_lanes;
  const unsigned long long bit = 1ull << bitpos;
  const double Lo = N.drv_length[other];
  if (oe & bit) {
    const int e = N.link_end_lane[k];
    const double2 tl_ = S.tail[IDX(e)];
    double td = tl_.x;
    int ttag = S.tail_tag[IDX(e)];
    int tt = tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(ttag));
    double tlen = N.tmpl[(size_t)tt * TSC_DT_STRIDE + TSC_T_LEN];
    double crossDistance = Lo - xo;
    double vehDistance = td - tlen;
    if (crossDistance + vehDistance < 0) {
      f.d2 = -(td + crossDistance);
      f.speed = tl_.y;
      f.vid = TSC_TAG_VID(ttag);
      f.tmpl = tt;
    }
  }
  if (f.vid < 0 && (ov & bit)) {
    int n = S.cnt[IDX(other)] & TSC_CNT_MASK;
    const int sb = N.drv_slot_base[other];
    for (int j = 0; j < n; j++) {
      const double2 vj = S.ds[DIDX(sb + j)];
      double dj = vj.x;
      int tg = S.tag[SIDX(sb + j)];
      int tt = tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(tg));
      double tlen = N.tmpl[(size_t)tt * TSC_DT_STRIDE + TSC_T_LEN];
      if (dj > xo && !(dj - xo - tlen <= 0)) continue; // rear already past the cross: next vehicle
      f.d2 = xo - dj;
      f.speed = vj.y;
      f.vid = TSC_TAG_VID(tg);
      f.ell = S.ab[SIDX(sb + j)].x;
      f.tmpl = tt;
      break;
    }
  }
  if (f.vid < 0 && (os_green & bit)) {
    const int sl = N.link_start_lane[k];
    const int s0 = N.drv_slot_base[sl];
    const double2 v0 = S.ds[DIDX(s0)];
    double ds = v0.x;
    int tg = S.tag[SIDX(s0)];
    f.d2 = (N.drv_length[sl] - ds) + xo;
    f.speed = v0.y;
    f.vid = TSC_TAG_VID(tg);
    f.tmpl = tmpl_of<SINGLE>(N, vbase, f.vid);
  }
  return f;
}

// ------------------------------------------------- car-following / cross helpers of k_exec
// Cross::canPass
__device__ __forceinline__ bool cross_can_pass(const DevNet &N, const DevState &S, int r, double speed, const Tm &t,
                                               int vid, int ell, int t1, int t2, double d1, const Foe &f,
                                               double interval) {
  Tm tf = load_tm(N, f.tmpl);
  double d2 = f.d2;
  if (!can_yield(speed, t, d1)) return true;
  int yield = 0;
  if (!can_yield(f.speed, tf, d2)) yield = 1;
  if (yield == 0) {
    double tgt1 = (t1 != TSC_LINK_STRAIGHT) ? t.turnSpeed : t.maxSpeed;
    double tgt2 = (t2 != TSC_LINK_STRAIGHT) ? tf.turnSpeed : tf.maxSpeed;
    if (t1 > t2) {
      yield = -1;
    } else if (t1 < t2) {
      if (d2 > 0) {
        int foeSteps = reach_steps(f.speed, d2, tgt2, tf.maxPosAcc, interval);
        int mySteps = reach_steps(speed, d1, tgt1, t.maxPosAcc, interval);
        if (foeSteps > mySteps) yield = -1;
      } else {
        if (d2 + tf.len < 0) yield = -1;
      }
      if (yield == 0) yield = 1;
    } else {
      if (d2 > 0) {
        int foeSteps = reach_steps(f.speed, d2, tgt2, tf.maxPosAcc, interval);
        int mySteps = reach_steps(speed, d1, tgt1, t.maxPosAcc, interval);
        if (foeSteps > mySteps) {
          yield = -1;
        } else if (foeSteps < mySteps) {
          yield = 1;
        } else {
          if (ell == f.ell) {
            if (d1 == d2) yield = tsc_goes_first(N, N.set_veh_ptr[S.set[r]], vid, f.vid) ? -1 : 1;
            else yield = d1 < d2 ? -1 : 1;
          } else {
            yield = ell < f.ell ? -1 : 1;
          }
        }
      } else {
        yield = d2 + tf.len < 0 ? -1 : 1;
      }
    }
  }
  if (yield == 1 && !(S.R < 0)) {
    // blocker-chain cycle check (Floyd) over the committed blockers
    int fast = f.vid, slow = f.vid;
    for (int guard = 0; guard < 4096; guard++) {
      if (fast < 0) break;
      int bf = S.bbv[IDX(fast)];
      if (bf < 0) break;
      slow = S.bbv[IDX(slow)];
      fast = S.bbv[IDX(bf)];
      if (slow == fast) {
        yield = -1;
        break;
      }
    }
  }
  return yield == -1;
}

// Blocker word of a slot: low 24 bits = blocker vid + 1 (0 = none); bit 30 = the
// vehicle had a blocker one step earlier (so the entry of the other ping-pong
// half of blocker_by_vid is refreshed once more after a blocker disappears).
#define TSC_BLK_VID(w) ((int)((w) & 0xffffff) - 1)
#define TSC_BLK_OLDER(w) (((w) >> 30) & 1)

// Resolve the cross conflicts of one vehicle on / approaching lane-link `link`
// (wave-uniform): first cross ahead, in distance order, that it may not pass.
template <bool SINGLE>
__device__ __forceinline__ void cross_scan(const DevNet &N, const DevState &S, int r, int vbase, const ListRec &lr,
                                           bool mine, double speed, const Tm &t, int vid, int ell, double dts,
                                           unsigned long long oe, unsigned long long ov, unsigned long long os_green,
                                           double interval, double &s, int &nblk, int ablate = 0) {
  unsigned long long active = mine ? (lr.conflict & (oe | ov | os_green)) : 0ull;
  if (!__builtin_amdgcn_ballot_w64(active != 0)) return; // wave-uniform early out
  const int t1 = lr.link_type;
  bool stop = active == 0;
  const int cb = lr.cross_b, ce = lr.cross_e;
  CrossRec cur = N.cross_rec[cb]; // wave-uniform: one 32-byte scalar load per cross, fetched one ahead
  for (int e = cb; e < ce; e++) {
    const CrossRec nxt = N.cross_rec[e + 1];
    const bool want = !stop && cur.x >= dts && ((active >> (cur.meta & 255)) & 1ull);
    if (__builtin_amdgcn_ballot_w64(want) && !TSC_ABLP(ablate, 128)) {
      if (want) {
        Foe f = find_foe<SINGLE>(N, S, r, vbase, cur.other, cur.meta & 255, cur.xo, oe, ov, os_green);
        if (f.vid >= 0 && !TSC_ABLP(ablate, 64)) {
          double d1 = cur.x - dts;
          if (!cross_can_pass(N, S, r, speed, t, vid, ell, t1, cur.meta >> 8, d1, f, interval)) {
            s = min2(s, stop_before_speed(speed, t, cur.x - dts - t.yieldDist, interval));
            nblk = f.vid;
            stop = true;
          }
        }
      }
    }
    cur = nxt;
  }
}

// --------------------------------------------------------- decision stage: k_enum / k_exec
// (1) k_enum -- enumeration: one wavefront = 64 replicas of ONE drivable; lane l walks the
//     slots of ITS replica (coalesced loads along the replica axis, wave-uniform topology,
//     next row in flight) and classifies every vehicle:
//       - lane vehicle stopped in a stopped queue, lane head standing at a red light
//         -> result known without arithmetic (v = 0, position unchanged), stored at once;
//       - plain follower (not the head, not within the approach distance of the lane
//         end)                                    -> the lane's follower list;
//       - head / near the lane end / on a lane-link -> the list of its next lane-link;
// (2) k_exec -- execution: 64 work items at a time, one per lane -- whatever replica and
//     slot they came from -- run the car-following / intersection logic with all lanes busy.
// Replicas hold very different numbers of moving vehicles, so walking slots in lock-step
// leaves most lanes idle (23 % of VALU lanes active in the v1 kernel); the engine's
// update is Jacobi (every decision reads only committed state), so vehicles can be
// executed in any grouping.
#define TSC_EU 2    /* slots enumerated per turn in k_enum */
#ifndef TSC_EXEC_OCC
#define TSC_EXEC_OCC 4 /* wavefronts per SIMD k_exec is compiled for (5: 102 registers -- measured) */
#endif
#ifndef TSC_ENUM_LB
#define TSC_ENUM_LB 8 /* lane-links per wavefront of k_enum */
#endif
#define TSC_MAXQ 4  /* out-links of a lane (checked at create): one work list per out-link */
#define TSC_NLIST (TSC_MAXQ + 1) /* work lists per lane: one per out-link + plain followers */
// work item: (replica << 13) | (slot << 4) | out-links-empty << 3 | route-ends-here << 2 |
//            has-blocker-word << 1 | is-tail
// (flags from k_enum's coalesced walk that spare k_exec scattered lines: aux is the list's
//  lane-link, or -1; an empty blocker word and empty out-links need not be fetched)
#define TSC_ITEM(rep, slot, oempty, ends, blk, tail)                                                          \
  ((unsigned int)(((unsigned int)(rep) << 13) | ((slot) << 4) | ((oempty) ? 8u : 0u) | ((ends) ? 4u : 0u) | \
                  ((blk) ? 2u : 0u) | ((tail) ? 1u : 0u)))
#define TSC_ITEM_TAIL(m) ((m) & 1u)
#define TSC_ITEM_BLK(m) (((m) >> 1) & 1u)
#define TSC_ITEM_ENDS(m) (((m) >> 2) & 1u)
#define TSC_ITEM_OEMPTY(m) (((m) >> 3) & 1u)
#define TSC_ITEM_REP(m) ((int)((m) >> 13))
#define TSC_ITEM_SLOT(m) ((int)(((m) >> 4) & 511u))

// ---- a plain follower (item: replica r, slot it >= 1)
template <bool SINGLE>
__device__ __forceinline__ void exec_follow(const DevNet &N, const DevState &S, const ListRec &lr, double interval,
                                            bool valid, int r, int it, bool is_tail, bool has_blk) {
  const int vbase = (SINGLE || !valid) ? 0 : N.set_veh_ptr[S.set[r]];
  const int d = lr.d, sb = lr.sb;
  const double L = lr.L, drvMax = lr.drvMax;
  const double2 *__restrict__ g_ds = S.ds;
  double2 *__restrict__ o_ds = S.nds;
  if (valid) {
        const double2 own = g_ds[DIDX(sb + it)];
        const double dist = own.x, speed = own.y;
        const double2 lead = g_ds[DIDX(sb + it - 1)];
        const double pd = lead.x, vL = lead.y;
        // the blocker word is almost always empty: k_enum says in the item whether to fetch it
        const int blkw = has_blk ? S.ab[SIDX(sb + it)].y : 0;
        int tid = 0, ptid = 0;
        if (!SINGLE) {
          tid = N.veh_tmpl[vbase + TSC_TAG_VID(S.tag[SIDX(sb + it)])];
          ptid = N.veh_tmpl[vbase + TSC_TAG_VID(S.tag[SIDX(sb + it - 1)])];
        }
        const Tm t = load_tm(N, tid);
        const Tm tl = load_tm(N, ptid);
        const double gap = pd - tl.len - dist;
        double v = t.maxSpeed;
        v = min2(v, speed + t.maxPosAcc * interval);
        v = min2(v, drvMax);
        v = min2(v, car_follow_speed(speed, t, vL, tl, gap, interval));
        v = max2(v, speed - t.maxNegAcc * interval);
        double deltaDis;
        if (v < 0) {
          deltaDis = 0.5 * speed * speed / t.maxNegAcc;
          v = 0;
        } else {
          deltaDis = (speed + v) * interval / 2;
        }
        if (blkw != 0) { // let a blocker from the previous intersection expire
          S.ab[SIDX(sb + it)].y = (TSC_BLK_VID(blkw) >= 0) ? (1 << 30) : 0;
          S.nbbv[IDX(TSC_TAG_VID(S.tag[SIDX(sb + it)]))] = -1;
        }
        const double nd = deltaDis + dist;
        if (nd > L) atomicAdd(&S.errors[3], 1); // cannot happen: more than the approach distance away
        o_ds[DIDX(sb + it)] = make_double2(nd, v);
        if (is_tail) {
          S.ntail[IDX(d)] = make_double2(nd, v);
        }
      }

}

// ---- head / near the lane end / on a lane-link (item: replica rl of the block, slot it);
//      lanes: `pick` = index of the out-link all items of this batch are heading for
template <bool SINGLE, bool IS_LINK>
__device__ __forceinline__ void exec_heavy(const DevNet &N, const DevState &S, const DevCfg &C, const ListRec &lr,
                                           bool valid, int r, int it, bool is_tail, bool route_ends, bool has_blk,
                                           bool out_empty) {
  const int d = lr.d, sb = lr.sb, ag = lr.ag;
  const double L = lr.L, drvMax = lr.drvMax, interval = C.interval;
  const int vbase = (SINGLE || !valid) ? 0 : N.set_veh_ptr[S.set[r]];
  const double2 *__restrict__ g_ds = S.ds;
  double2 *__restrict__ o_ds = S.nds;
      // the lane-link this batch deals with (wave-uniform); lanes: out-link `pick`
      const int link = lr.link;
      const int el = lr.el;
      // signal phase and cross-candidate masks of the item's replica: loaded with the item
      int raw = 0;
      unsigned long long oe = 0, ov = 0, os_green = 0;
      if (valid && ag >= 0 && !TSC_ABL(32768)) {
        const AgentRec ar = S.arec[IDX(ag)]; // one 32-byte record: one line per item instead of four
        raw = ar.raw;
        oe = ar.oe;
        ov = ar.ov;
        os_green = ar.os;
      }
      // ---- ONE load group: the item and its list predecessor, the header of the own
      //      link's end lane (look-ahead, canEnter) and, for head vehicles, the tails of all
      //      out-links (look-ahead).  Addresses depend only on (replica, slot) and
      //      wave-uniform table entries, so everything is in flight together.
      double dist = 0, speed = 0, pd = 0, ps = 0;
      int tag = 0, aux = -1, blkw = 0, ptag = 0;
      int l_cnt[TSC_MAXQ], l_tt[TSC_MAXQ];
      double l_td[TSC_MAXQ], l_ts[TSC_MAXQ];
      int e_hdr = 0, e_tt = 0;
      double e_td = 0, e_ts = 0;
      if (valid) {
        const double2 own = g_ds[DIDX(sb + it)];
        dist = own.x;
        speed = own.y;
        // the tag (vehicle id) of a single-template lane vehicle is only needed by the cross scan,
        // a blocker change or a hand-off: fetched there, not for every item
        if (!SINGLE || IS_LINK) tag = S.tag[SIDX(sb + it)];
        if (IS_LINK || has_blk) { // lane-links: aux is the entry time; else only a non-empty blocker word needs the line
          const int2 om = S.ab[SIDX(sb + it)];
          aux = om.x;
          blkw = om.y;
        }
        if (!IS_LINK) aux = route_ends ? -1 : lr.link; // the list a lane vehicle is in says where it is heading
        if (it > 0 && !TSC_ABL(131072)) {
          const double2 lead = g_ds[DIDX(sb + it - 1)];
          pd = lead.x;
          ps = lead.y;
          if (!SINGLE) ptag = S.tag[SIDX(sb + it - 1)];
        }
      }
      if (valid && (IS_LINK ? it == 0 : link >= 0) && !TSC_ABL(65536)) {
        e_hdr = S.cnt[IDX(el)];
        const double2 et_ = S.tail[IDX(el)];
        e_td = et_.x;
        e_ts = et_.y;
        if (!SINGLE) e_tt = S.tail_tag[IDX(el)];
      }
      if (!IS_LINK) {
        const LaneMaskRec lm_ = N.lane_mask[d]; // (wave-uniform: the lane's out-links with their local indices)
#pragma unroll
        for (int qq = 0; qq < TSC_MAXQ; qq++) {
          l_cnt[qq] = 0;
          l_tt[qq] = 0;
          l_td[qq] = 0;
          l_ts[qq] = 0;
          // (round 4: only the out-links that hold a vehicle -- the intersection's maintained mask says which, and it is
          //  in registers already: their counts are not read at all, the empty ones' tails neither)
          bool occ = false;
          if (lr.olk[qq] >= 0 && valid && it == 0 && !out_empty) {
            unsigned int loc = 64u;
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
              if (lm_.oid[q2] == lr.olk[qq]) loc = (lm_.oloc >> (8 * q2)) & 63u;
            occ = loc < 64u && ((ov >> loc) & 1ull) != 0;
          }
          if (occ) {
            l_cnt[qq] = 1;
            const double2 lt_ = S.tail[IDX(lr.olk[qq])];
            l_td[qq] = lt_.x;
            l_ts[qq] = lt_.y;
            if (!SINGLE) l_tt[qq] = S.tail_tag[IDX(lr.olk[qq])];
          }
        }
      }
      const bool has_link = IS_LINK ? false : (valid && aux >= 0); // then aux == link (bucketed)
      // a head-of-lane candidate only counts while its link is green (engine: notifyCross)
      if (ag >= 0) os_green &= N.agent_phase_avail[(size_t)ag * 32 + raw];
      const double my_len = lr.my_len;
      bool have_tag = valid && (!SINGLE || IS_LINK);
      int vid = TSC_TAG_VID(tag);
      const int tid = tmpl_of<SINGLE>(N, vbase, vid);
      const Tm t = load_tm(N, tid);
      const int e_tmpl = SINGLE ? 0 : tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(e_tt));
      const double e_len = N.tmpl[(size_t)e_tmpl * TSC_DT_STRIDE + TSC_T_LEN];

      // ---- leader and gap (engine: updateLeaderAndGap)
      bool has_leader = false;
      double gap = 0, vL = 0;
      int l_tmpl = 0;
      if (valid && it > 0) {
        has_leader = true;
        l_tmpl = SINGLE ? 0 : tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(ptag));
        gap = pd - N.tmpl[(size_t)l_tmpl * TSC_DT_STRIDE + TSC_T_LEN] - dist;
        vL = ps;
      } else if (valid) {
        double dis = L - dist;
        const double limit = t.limit;
        const int ne = (e_hdr & TSC_CNT_MASK) - ((e_hdr & TSC_NEW_BIT) ? 1 : 0); // tail before this step's insertion
        if (IS_LINK) {
          if (ne > 0) {
            has_leader = true;
            gap = dis + e_td - e_len;
            vL = e_ts;
            l_tmpl = e_tmpl;
          }
        } else if (has_link) {
          // all lane-links leaving this lane may overlap: nearest tail wins
#pragma unroll
          for (int qq = 0; qq < TSC_MAXQ; qq++) {
            if (lr.olk[qq] >= 0 && l_cnt[qq] > 0) {
              int tt = SINGLE ? 0 : tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(l_tt[qq]));
              double cg = dis + l_td[qq] - N.tmpl[(size_t)tt * TSC_DT_STRIDE + TSC_T_LEN];
              if (!has_leader || cg < gap) {
                has_leader = true;
                gap = cg;
                vL = l_ts[qq];
                l_tmpl = tt;
              }
            }
          }
          if (!has_leader) {
            dis += my_len;
            if (!(dis > limit) && ne > 0) {
              has_leader = true;
              gap = dis + e_td - e_len;
              vL = e_ts;
              l_tmpl = e_tmpl;
              // beyond: a whole lane further is past the look-ahead limit (checked at create)
            }
          }
        }
      }

      // ---- speed choice (engine: getNextSpeed)
      double v = t.maxSpeed;
      v = min2(v, speed + t.maxPosAcc * interval);
      v = min2(v, drvMax);
      if (has_leader) {
        Tm tl = load_tm(N, l_tmpl);
        v = min2(v, car_follow_speed(speed, t, vL, tl, gap, interval));
      } else {
        v = min2(v, t.maxSpeed);
      }
      int nblk = -1;
      if (IS_LINK) {
        double s = t.maxSpeed;
        if (!TSC_ABL(16))
          cross_scan<SINGLE>(N, S, r, vbase, lr, valid, speed, t, vid, aux, dist, oe, ov, os_green, interval, s, nblk,
                             TSC_ABLV);
        v = min2(v, s);
      } else {
        const bool related = has_link && (L - dist) <= t.approach && !TSC_ABL(1);
        if (__builtin_amdgcn_ballot_w64(related)) {
          double s = t.maxSpeed;
          bool done = !related;
          if (related) {
            const bool avail = (lr.avail_mask >> raw) & 1u;
            bool can_enter;
            if ((e_hdr & TSC_CNT_MASK) == 0) can_enter = true;
            else if (e_hdr & TSC_NEW_BIT) can_enter = false; // tail is a vehicle inserted this step: dist 0, speed 0
            else can_enter = e_td > e_len + t.len || e_ts >= 2;
            if (!avail || !can_enter) {
              if (min_brake_distance(speed, t) > L - dist) {
                // cannot brake before the stop line
              } else {
                s = min2(s, stop_before_speed(speed, t, L - dist, interval));
                done = true;
              }
            }
            if (!done && lr.link_type != TSC_LINK_STRAIGHT) s = min2(s, t.turnSpeed);
          }
          // A non-head vehicle whose first cross lies beyond its braking distance after one
          // more acceleration step cannot be slowed by any cross (stop_before_speed would
          // return speed + maxPosAcc*dt, already a bound of v), and its blocker is never read
          // (see the enumeration note): the scan is skipped.  Heads are always scanned.
          bool scan = !done;
          if (scan && it > 0 && link >= 0) {
            if (lr.cross_b < lr.cross_e) {
              const double nextSpeed = speed + t.maxPosAcc * interval;
              const double bd = (speed + nextSpeed) * interval / 2 + (nextSpeed * nextSpeed / t.usualNegAcc / 2);
              if (bd < lr.first_cross_x + (L - dist) - t.yieldDist) scan = false;
            } else scan = false;
          }
          if (scan && !have_tag) {
            tag = S.tag[SIDX(sb + it)];
            vid = TSC_TAG_VID(tag);
            have_tag = true;
          }
          if (!TSC_ABL(16))
            cross_scan<SINGLE>(N, S, r, vbase, lr, scan, speed, t, vid, 0x7fffffff, -(L - dist), oe, ov, os_green,
                               interval, s, nblk, TSC_ABLV);
          if (related) v = min2(v, s);
        }
      }
      v = max2(v, speed - t.maxNegAcc * interval);
      double deltaDis;
      if (v < 0) {
        deltaDis = 0.5 * speed * speed / t.maxNegAcc;
        v = 0;
      } else {
        deltaDis = (speed + v) * interval / 2;
      }
      if (valid) {
        // ---- blocker bookkeeping: slot word in place, vid-indexed entry in the other half
        int old_b = TSC_BLK_VID(blkw);
        int new_w = ((nblk + 1) & 0xffffff) | ((old_b >= 0) ? (1 << 30) : 0);
        if (new_w != blkw) S.ab[SIDX(sb + it)].y = new_w;
        const bool touch_bbv = nblk >= 0 || old_b >= 0 || TSC_BLK_OLDER(blkw);
        // ---- move (engine: setDeltaDistance)
        double nd = deltaDis + dist;
        if (!have_tag && (touch_bbv || nd > L)) {
          tag = S.tag[SIDX(sb + it)];
          vid = TSC_TAG_VID(tag);
        }
        if (touch_bbv) S.nbbv[IDX(vid)] = nblk;
        if (nd > L) {
          nd -= L;
          int ko = TSC_OB_OF(atomicAdd(&S.cnt[IDX(d)], TSC_OB_ONE));
          if (ko < TSC_OB_MAX) {
            size_t ox = ((size_t)d * TSC_OB_MAX + ko) * (size_t)S.RS + (size_t)r;
            ObxRec rec; // one 32-byte record: one line written here, one read by k_commit
            rec.dist = nd;
            rec.speed = v;
            rec.tag = tag;
            rec.aux = aux; // lane vehicle with aux < 0: leaves the network
            rec.blk = new_w;
            rec.slot = it;
            S.obx[ox] = rec;
          } else atomicAdd(&S.errors[1], 1);
          // ... and tell the drivable it goes to (round 4: k_commit's "is anything arriving?" was a look at every feeder's
          // outbox count for every drivable and replica in every sub-step: four rows per lane, one per lane-link)
          const int tgt = IS_LINK ? el : aux;
          if (tgt >= 0) atomicOr((unsigned int *)&S.cnt[IDX(tgt)], (unsigned int)TSC_ARR_BIT);
        }
        o_ds[DIDX(sb + it)] = make_double2(nd, v);
        if (is_tail) { // next tail record of a drivable that stays quiet (k_commit redoes it otherwise)
          S.ntail[IDX(d)] = make_double2(nd, v);
        }
      }

}

// Work lists: one per (lane, out-link), one per lane for its plain followers, one per
// lane-link -- each spanning ALL replicas, so that execution batches are full whatever
// the per-replica occupancy.  A k_enum wave stages its items in LDS and appends them 64
// at a time (one atomicAdd and one coalesced 256-byte store per flush); the order inside
// a list is irrelevant (Jacobi update).
//   list index: lane l, queue q in [0, TSC_NLIST)  ->  l * TSC_NLIST + q
//               lane-link k                          ->  n_lanes * TSC_NLIST + k
//   storage of list i: wl_items[wl_off[i] .. wl_off[i+1]), sized for every slot of the
//   drivable in every replica, so it cannot overflow.
#define TSC_STAGE 128
__device__ __forceinline__ void wl_flush(const DevState &S, int list, unsigned int *stage, int n) {
  int base = 0;
  if (threadIdx.x == 0) base = atomicAdd(&S.wl_cnt[list], n);
  base = __builtin_amdgcn_readfirstlane(base);
  if ((int)threadIdx.x < n) S.wl_items[S.wl_rec[list].item_off + base + threadIdx.x] = stage[threadIdx.x];
}
// one wave per workgroup: __syncthreads() is a wave-local LDS fence here
__device__ __forceinline__ void wl_put(const DevState &S, int list, unsigned int *stage, int &qn, bool put,
                                       unsigned int item) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(put);
  if (!m) return;
  const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
  if (put) stage[qn + rank] = item;
  qn += __builtin_popcountll(m);
  if (qn >= 64) {
    __syncthreads();
    wl_flush(S, list, stage, 64);
    const unsigned int rest = stage[64 + threadIdx.x];
    __syncthreads();
    stage[threadIdx.x] = rest;
    qn -= 64;
  }
}

// ------------------------------------------------------------------ k_enum
template <bool SINGLE>
__global__ __launch_bounds__(64, 6) void k_enum(DevNet N, DevState S, DevCfg C, int n_rb, int n_rows) {
  __shared__ unsigned int stage[TSC_NLIST][TSC_STAGE];
  int rb, row;
  if (!map_block(n_rb, n_rows, rb, row)) return;
  row = __builtin_amdgcn_readfirstlane(row);
  const bool IS_LINK = row >= N.n_lanes; // rows: the lanes, then the lane-links in groups of TSC_ENUM_LB
  const int lane = threadIdx.x;
  const int r = rb * 64 + lane;
  const bool own_valid = r < S.R && !TSC_PARKED(S, r);
  if (row == 0 && own_valid) S.t[r] += 1; // the step index being executed from here on (the first kernel of a sub-step)
  if (IS_LINK) {
    // Round 4: a wavefront takes TSC_ENUM_LB lane-links, their counts requested together.  Three quarters of the rows
    // of a grid are lane-links and nearly all of them are empty in all 64 replicas: one wavefront per lane-link was
    // 590 k wavefronts per launch on the 16x16 grid that each waited for ONE round trip to memory and left.
    const int k0 = (row - N.n_lanes) * TSC_ENUM_LB;
    int nn[TSC_ENUM_LB];
#pragma unroll
    for (int u = 0; u < TSC_ENUM_LB; u++) {
      nn[u] = 0;
      if (own_valid && k0 + u < N.n_links) {
        const int hdr = S.cnt[IDX(N.n_lanes + k0 + u)];
        nn[u] = hdr & TSC_CNT_MASK;
        if (hdr & TSC_OB_BITS) S.cnt[IDX(N.n_lanes + k0 + u)] = hdr & ~TSC_OB_BITS; // outbox of this sub-step (the executors append with atomics)
      }
    }
#pragma unroll
    for (int u = 0; u < TSC_ENUM_LB; u++) {
      const int n = nn[u];
      if (!__builtin_amdgcn_ballot_w64(n > 0)) continue;
      // every vehicle on a lane-link takes the full path
      const int wl = N.n_lanes * TSC_NLIST + k0 + u;
      int qn = 0;
      for (int i = 0; __builtin_amdgcn_ballot_w64(i < n); i++) wl_put(S, wl, stage[0], qn, i < n, TSC_ITEM(r, i, 0, 0, 1, i == n - 1));
      if (qn) {
        __syncthreads();
        wl_flush(S, wl, stage[0], qn);
        __syncthreads(); // (the staging area is the next lane-link's)
      }
    }
    return;
  }
  const int d = row;
  const int wl0 = row * TSC_NLIST;
  // the drivable's constants in one 128-byte scalar load (the record of its first list)
  const ListRec lr0 = S.wl_rec[wl0];
  const int sb = lr0.sb;
  const double L = lr0.L;
  int n = 0;
  if (own_valid) {
    const int hdr = S.cnt[IDX(d)];
    n = hdr & TSC_CNT_MASK;
    if (hdr & TSC_OB_BITS) S.cnt[IDX(d)] = hdr & ~TSC_OB_BITS; // outbox of this sub-step (the executors append with atomics)
  }
  if (!__builtin_amdgcn_ballot_w64(n > 0)) return;
  int olk[TSC_MAXQ];
#pragma unroll
  for (int qq = 0; qq < TSC_MAXQ; qq++) olk[qq] = lr0.olk[qq];
  const double approach0 = N.tmpl[TSC_T_APPROACH], len0 = N.tmpl[TSC_T_LEN], mingap0 = N.tmpl[TSC_T_MINGAP];
  const double bd0 = brake_distance_after_accel(0.0, load_tm(N, 0), C.interval);
  // is any vehicle on a lane-link leaving this lane?  (look-ahead of a free-running head; loaded
  // with the first slots, not after them)
  // (round 4: from the intersection's "vehicles on the lane-link" mask, which k_commit keeps up to date and which comes
  //  with the phase in one 32-byte record -- not from the four out-links' header words: four rows less per lane)
  bool out_links_empty = true;
  // which out-links are green in this replica (bit q): a head stopped at a red light is settled here
  int green_bits = 0xf;
  if (lr0.ag >= 0 && own_valid) {
    const AgentRec ar_ = S.arec[IDX(lr0.ag)];
    out_links_empty = (ar_.ov & N.lane_mask[d].om) == 0ull;
    const int raw = ar_.raw;
    green_bits = 0;
#pragma unroll
    for (int qq = 0; qq < TSC_MAXQ; qq++)
      if (olk[qq] >= 0 && ((lr0.avail4[qq] >> raw) & 1u)) green_bits |= 1 << qq;
  }
  const double2 *__restrict__ g_ds = S.ds;
  double2 *__restrict__ o_ds = S.nds;
  const int vbase = (SINGLE || !own_valid) ? 0 : N.set_veh_ptr[S.set[r]];
  // ---- the standing block of this (lane, replica): slots (fa, fb] were settled in the
  //      previous sub-step with an empty blocker word, behind slot fa which was settled
  //      too.  Their committed state is therefore the one they had (speed 0, same place,
  //      same leader state) and BOTH ping-pong halves already hold it: they are settled
  //      again, without being read or written (the walk stays in slot lock-step across
  //      the 64 replicas -- row loads stay coalesced -- the lanes inside their block are
  //      just masked off).  Slot fa -- whose own leader may have moved -- is walked like
  //      any other; k_commit drops the block when slots shift.
  int fa = 0, fb = 0, fz = 0;
  if (own_valid && !TSC_ABL(8)) {
    fz = S.fz[IDX(d)];
    fa = fz & 0xffff;
    fb = fz >> 16;
    if (fb >= n) fb = fa; // cannot happen (slots only shift with a reset of the block)
  }
  const int skip = fb > fa ? fb - fa : 0;
  double f_dist = 0;
  int f_tmpl = 0;
  if (skip) {
    f_dist = g_ds[DIDX(sb + fb)].x;
    if (!SINGLE) f_tmpl = N.veh_tmpl[vbase + TSC_TAG_VID(S.tag[SIDX(sb + fb)])];
    if (fb == n - 1) { // the tail record follows the last vehicle even when it stands
      S.ntail[IDX(d)] = make_double2(f_dist, 0.0);
    }
  }
#define TSC_WALKED(sl) ((sl) < n && !((sl) > fa && (sl) <= fb && skip))
  double p_dist = 0, p_speed = 0;
  int p_tmpl = 0;
  // the first run of settled slots with an empty blocker word that reaches the old block
  // (or the first such run when there is none): next sub-step's block
  int run_a = -1, run_b = -1;
  bool run_final = false;
  int qn[TSC_NLIST];
#pragma unroll
  for (int q = 0; q < TSC_NLIST; q++) qn[q] = 0;
  // software pipeline: TSC_EU slots per turn, the next TSC_EU in flight
  double c_dist[TSC_EU], c_speed[TSC_EU];
  int c_blk[TSC_EU], c_tag[TSC_EU], c_aux[TSC_EU];
#pragma unroll
  for (int u = 0; u < TSC_EU; u++) {
    c_dist[u] = 0;
    c_speed[u] = 0;
    c_blk[u] = 0;
    c_tag[u] = 0;
    c_aux[u] = -1;
    if (TSC_WALKED(u)) {
      const int sl = u;
      const double2 v_ = g_ds[DIDX(sb + sl)];
      const int2 m_ = S.ab[SIDX(sb + sl)];
      c_dist[u] = v_.x;
      c_speed[u] = v_.y;
      c_blk[u] = m_.y;
      c_aux[u] = m_.x;
      if (!SINGLE) c_tag[u] = S.tag[SIDX(sb + sl)];
    }
  }
  for (int i = 0; __builtin_amdgcn_ballot_w64(i < n); i += TSC_EU) {
    double dist2[TSC_EU], speed2[TSC_EU];
    int blk2[TSC_EU], tag2[TSC_EU], aux2[TSC_EU];
#pragma unroll
    for (int u = 0; u < TSC_EU; u++) {
      dist2[u] = c_dist[u];
      speed2[u] = c_speed[u];
      blk2[u] = c_blk[u];
      tag2[u] = c_tag[u];
      aux2[u] = c_aux[u];
    }
#pragma unroll
    for (int u = 0; u < TSC_EU; u++)
      if (TSC_WALKED(i + TSC_EU + u)) {
        const int sl = i + TSC_EU + u;
        const double2 v_ = g_ds[DIDX(sb + sl)];
        const int2 m_ = S.ab[SIDX(sb + sl)];
        c_dist[u] = v_.x;
        c_speed[u] = v_.y;
        c_blk[u] = m_.y;
        c_aux[u] = m_.x;
        if (!SINGLE) c_tag[u] = S.tag[SIDX(sb + sl)];
      }
#pragma unroll
    for (int u = 0; u < TSC_EU; u++) {
      const int is = i + u;
      const bool active = TSC_WALKED(is);
      const double dist = dist2[u], speed = speed2[u];
      const int blkw = blk2[u];
      int tid = 0, bucket = -1;
      if (active) {
        if (skip && is == fb + 1) {
          // first slot behind the standing block: its leader is the block's last vehicle;
          // the block joins the run that reaches it (or starts one behind a moving slot fa)
          p_dist = f_dist;
          p_speed = 0.0;
          p_tmpl = f_tmpl;
          if (!run_final) {
            if (run_a < 0) run_a = fa + 1;
            run_b = fb;
          }
        }
        tid = SINGLE ? 0 : N.veh_tmpl[vbase + TSC_TAG_VID(tag2[u])];
        const double approach = SINGLE ? approach0 : N.tmpl[(size_t)tid * TSC_DT_STRIDE + TSC_T_APPROACH];
        const bool heavy = is == 0 || (L - dist) <= approach;
        bucket = TSC_MAXQ; // plain follower
        if (heavy) {
          bucket = 0;
#pragma unroll
          for (int qq = 1; qq < TSC_MAXQ; qq++)
            if (olk[qq] == aux2[u]) bucket = qq;
        }
        bool settled = false;
        if (u == 0 && is == 0 && speed == 0 && aux2[u] >= 0 && !((green_bits >> bucket) & 1) && !TSC_ABL(4096)) {
          // The head of a lane standing at a red light, too close to the stop line to take
          // one more acceleration step: the stop-line bound of getNextSpeed is 0 (its
          // link is not available and it can brake), no cross is scanned (so no blocker),
          // every other bound is clamped by it: v = 0 and the position is unchanged.
          // (stop_before_speed(0, rest) is 0 exactly when one more acceleration step could
          // not be braked off within `rest` > 0; at rest == 0 it is 0/0, left to k_exec.)
          const double rest = L - dist;
          const double bd = SINGLE ? bd0 : brake_distance_after_accel(0.0, load_tm(N, tid), C.interval);
          if (rest <= approach && rest > 0.0 && !(bd < rest)) settled = true;
        }
        if (is > 0 && speed == 0 && p_speed == 0) {
          // A lane vehicle stopped behind a stopped leader within minGap: both
          // no-collision bounds are negative whatever the signal or the crosses say, so
          // v = 0 and the position is unchanged -- no arithmetic needed.  (The blocker
          // the engine would note for it is never read: only lane heads, vehicles on
          // lane-links and fresh tails are ever registered at a cross.)
          const double plen = SINGLE ? len0 : N.tmpl[(size_t)p_tmpl * TSC_DT_STRIDE + TSC_T_LEN];
          const double mg = SINGLE ? mingap0 : N.tmpl[(size_t)tid * TSC_DT_STRIDE + TSC_T_MINGAP];
          if (p_dist - plen - dist <= mg) settled = true;
        }
        double out_d = dist, out_v = 0.0;
        bool finished = settled;
        if (u == 0 && is == 0 && !settled && !TSC_ABL(8192)) {
          // A lane head farther from the lane end than the look-ahead limit, with no vehicle
          // on any lane-link leaving the lane (or leaving the network at the lane end): no
          // leader within reach, not yet concerned by the signal or the crosses -- free
          // acceleration, three comparisons.  It cannot reach the lane end in this step.
          const Tm t = load_tm(N, tid);
          const double rest = L - dist;
          const double limit = t.limit;
          if (rest > limit && rest > t.approach && rest > 2 * t.maxSpeed * C.interval) {
            const bool clear = aux2[u] < 0 || out_links_empty;
            if (clear) {
              double v = t.maxSpeed;
              v = min2(v, speed + t.maxPosAcc * C.interval);
              v = min2(v, lr0.drvMax);
              v = min2(v, t.maxSpeed);
              v = max2(v, speed - t.maxNegAcc * C.interval);
              if (!(v < 0)) {
                out_v = v;
                out_d = (speed + v) * C.interval / 2 + dist;
                finished = true;
              }
            }
          }
        }
        if (finished) {
          bucket = -1;
          if (blkw != 0) { // let an earlier blocker entry expire
            S.ab[SIDX(sb + is)].y = (TSC_BLK_VID(blkw) >= 0) ? (1 << 30) : 0;
            S.nbbv[IDX(TSC_TAG_VID(S.tag[SIDX(sb + is)]))] = -1;
          }
          // (the first slot of a standing block was settled in the previous sub-step as well: the
          //  other half already holds this very record)
          if (!(settled && skip && is == fa)) o_ds[DIDX(sb + is)] = make_double2(out_d, out_v);
          if (is == n - 1) {
            S.ntail[IDX(d)] = make_double2(out_d, out_v);
          }
        }
        if (!run_final) {
          if (settled && blkw == 0) {
            if (run_a < 0) run_a = is;
            run_b = is;
          } else if (run_a >= 0 && (skip == 0 || is > fb)) {
            run_final = true; // the run ended behind the old block (or there was none)
          } else {
            run_a = -1; // a run that ended before reaching the old block
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(bucket >= 0)) {
        const unsigned int item = TSC_ITEM(r, is, out_links_empty, aux2[u] < 0, blkw != 0, is == n - 1);
#pragma unroll
        for (int q = 0; q < TSC_NLIST; q++) {
          if (q > 0 && q < TSC_MAXQ && olk[q] < 0) continue; // no such out-link (wave-uniform)
          if (!TSC_ABL(256)) wl_put(S, wl0 + q, stage[q], qn[q], bucket == q, item);
        }
      }
      p_dist = dist;
      p_speed = speed;
      p_tmpl = tid;
    }
  }
#undef TSC_WALKED
  if (own_valid) {
    // a block whose walk ended inside it (fb == n - 1) is joined here
    if (skip && fb == n - 1 && !run_final) {
      if (run_a < 0) run_a = fa + 1;
      run_b = fb;
    }
    const int nfz = (run_a >= 0 && run_b > run_a) ? (run_a | (run_b << 16)) : 0;
    if (nfz != fz) S.fz[IDX(d)] = nfz;
  }
  __syncthreads();
  // the five reservations are issued together (one round trip), then the five stores
  int fbase[TSC_NLIST];
#pragma unroll
  for (int q = 0; q < TSC_NLIST; q++) {
    fbase[q] = 0;
    if (qn[q] && threadIdx.x == 0) fbase[q] = atomicAdd(&S.wl_cnt[wl0 + q], qn[q]);
  }
#pragma unroll
  for (int q = 0; q < TSC_NLIST; q++)
    if (qn[q]) {
      const int base = __builtin_amdgcn_readfirstlane(fbase[q]);
      if ((int)threadIdx.x < qn[q])
        S.wl_items[S.wl_rec[wl0 + q].item_off + base + threadIdx.x] = stage[q][threadIdx.x];
    }
}

// ------------------------------------------------------------------ k_plan
// Turns the lists' item counts into batch descriptors (list, batch) appended to the batch table.  One workgroup per
// PLAN_LISTS lists: a block scan of their batch counts, ONE atomic add on the table's fill word per workgroup (one per
// list was 25 k adds to one address on the 16x16 grid: the whole 137 us of the kernel), then the workgroup's descriptors
// are written cooperatively, descriptor t by thread t mod PLAN_LISTS (its list found by bisection of the prefix sums in LDS).
#define PLAN_LISTS 256
__global__ __launch_bounds__(PLAN_LISTS) void k_plan(DevNet N, DevState S) {
  __shared__ int pre[PLAN_LISTS + 1];
  __shared__ int wsum[PLAN_LISTS / 64];
  __shared__ int base_s;
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int list0 = (int)blockIdx.x * PLAN_LISTS, list = list0 + tid;
  const int nb = list < S.n_lists ? (S.wl_cnt[list] + 63) >> 6 : 0;
  int x = nb; // inclusive scan inside the wavefront
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int i = 0; i < PLAN_LISTS / 64; i++) {
    if (i < w) off += wsum[i];
    total += wsum[i];
  }
  pre[tid] = off + x - nb;
  if (tid == 0) {
    pre[PLAN_LISTS] = total;
    base_s = total ? atomicAdd(&S.wl_nb[0], total) : 0;
  }
  __syncthreads();
  const int base = base_s;
  for (int t = tid; t < total; t += PLAN_LISTS) {
    int lo = 0; // the last list whose first descriptor is not beyond t
#pragma unroll
    for (int st = PLAN_LISTS / 2; st; st >>= 1)
      if (pre[lo + st] <= t) lo += st;
    S.wl_desc[base + t] = make_int2(list0 + lo, t - pre[lo]);
  }
}

// ------------------------------------------------------------------ k_exec
// Persistent grid over the batch table: workgroup w takes batches w, w + G, ...
template <bool SINGLE>
__global__ __launch_bounds__(64, TSC_EXEC_OCC) void k_exec(DevNet N, DevState S, DevCfg C) {
  const int lane = threadIdx.x;
  const int nb = S.wl_nb[0];
  const int2 *desc = S.wl_desc;
  int b = blockIdx.x;
  if (b >= nb) return;
  // the descriptor and the items of the next batch are fetched while this one executes
  int2 de = desc[b];
  unsigned int item = 0;
  int total = 0;
  {
    const int list = __builtin_amdgcn_readfirstlane(de.x), j = __builtin_amdgcn_readfirstlane(de.y);
    total = S.wl_cnt[list];
    if (j * 64 + lane < total) item = S.wl_items[S.wl_rec[list].item_off + j * 64 + lane];
  }
  while (true) {
    const int list = __builtin_amdgcn_readfirstlane(de.x);
    const int j = __builtin_amdgcn_readfirstlane(de.y);
    const ListRec lr = S.wl_rec[list];
    const bool valid = j * 64 + lane < total;
    const unsigned int it_cur = item;
    const int b2 = b + (int)gridDim.x;
    if (b2 < nb) {
      de = desc[b2];
      const int list2 = __builtin_amdgcn_readfirstlane(de.x), j2 = __builtin_amdgcn_readfirstlane(de.y);
      total = S.wl_cnt[list2];
      item = (j2 * 64 + lane < total) ? S.wl_items[S.wl_rec[list2].item_off + j2 * 64 + lane] : 0u;
    }
    const bool skip = !lr.is_link && ((TSC_ABL(1024) && lr.pick == TSC_MAXQ) || (TSC_ABL(2048) && lr.pick != TSC_MAXQ));
    const int r = TSC_ITEM_REP(it_cur);
    if (skip) {
    } else if (lr.is_link) {
      exec_heavy<SINGLE, true>(N, S, C, lr, valid, r, TSC_ITEM_SLOT(it_cur), valid && TSC_ITEM_TAIL(it_cur), false, true, false);
    } else if (lr.pick == TSC_MAXQ) {
      exec_follow<SINGLE>(N, S, lr, C.interval, valid, r, valid ? TSC_ITEM_SLOT(it_cur) : 1, valid && TSC_ITEM_TAIL(it_cur),
                          valid && TSC_ITEM_BLK(it_cur));
    } else {
      exec_heavy<SINGLE, false>(N, S, C, lr, valid, r, TSC_ITEM_SLOT(it_cur), valid && TSC_ITEM_TAIL(it_cur),
                                TSC_ITEM_ENDS(it_cur), TSC_ITEM_BLK(it_cur), TSC_ITEM_OEMPTY(it_cur));
    }
    if (b2 >= nb) break;
    b = b2;
  }
}

// ---------------------------------------------------------------- k_commit
template <bool SINGLE>
__device__ __forceinline__ void spawn_if_due(const DevNet &N, const DevState &S, double2 *q_ds,
                                             int r, int d, int sb, int t_next, int &w, bool have_tail,
                                             double tail_dist, int tail_tmpl, int &has_new) {
  int set = S.set[r];
  int qb = N.spawn_ptr[set * N.n_lanes + d], qe = N.spawn_ptr[set * N.n_lanes + d + 1];
  if (qb == qe) return;
  int c = S.spawn_cur[IDX(d)];
  if (qb + c >= qe) return;
  int vbase = N.set_veh_ptr[set];
  int vid = N.spawn_vid[qb + c];
  if (N.veh_start[vbase + vid] > t_next) return;
  int tid = tmpl_of<SINGLE>(N, vbase, vid);
  if (have_tail) {
    double tlen = N.tmpl[(size_t)tail_tmpl * TSC_DT_STRIDE + TSC_T_LEN];
    double mg = N.tmpl[(size_t)tid * TSC_DT_STRIDE + TSC_T_MINGAP];
    if (!(tail_dist > tlen + mg)) return;
  }
  if (w >= N.drv_cap[d]) {
    atomicAdd(&S.errors[0], 1);
    return;
  }
  const int2 rr_ = N.veh_rrange[vbase + vid];
  int rb_ = rr_.x, re_ = rr_.y;
  q_ds[DIDX(sb + w)] = make_double2(0.0, 0.0);
  S.tag[SIDX(sb + w)] = TSC_MAKE_TAG(vid, 0);
  S.ab[SIDX(sb + w)] = make_int2((rb_ + 1 < re_) ? N.route_drv[rb_ + 1] : -1, 0);
  S.bbv[IDX(vid)] = -1;
  S.nbbv[IDX(vid)] = -1;
  S.spawn_cur[IDX(d)] = c + 1;
  w++;
  has_new = 1;
}

template <bool SINGLE>
__global__ __launch_bounds__(256) void k_commit(DevNet N, DevState S, DevCfg C, int n_rb, int n_dblk) {
  int rb, dblk;
  if (!map_block(n_rb, n_dblk, rb, dblk)) return;
  int d = __builtin_amdgcn_readfirstlane(dblk * (int)blockDim.y + (int)threadIdx.y);
  int r = rb * 64 + threadIdx.x;
  if (d >= N.n_drv || r >= S.R || TSC_PARKED(S, r)) return;
  const bool is_link = d >= N.n_lanes;
  const int sb = N.drv_slot_base[d];
  const int cap = N.drv_cap[d];
  // the half k_enum / k_exec just wrote: it becomes the committed state after this kernel
  double2 *__restrict__ q_ds = S.nds;
  // ---- everything the quiet test needs is loaded up front, in one round trip
  int fb, fe;
  if (is_link) {
    fb = 0;
    fe = 1;
  } else {
    fb = N.lane_in_ptr[d];
    fe = N.lane_in_ptr[d + 1];
  }
  const int hdr = S.cnt[IDX(d)];
  const int o_raw = TSC_OB_OF(hdr);
  const int set_r = S.set[r];
  const int t_now = S.t[r];
  // (round 4: the feeders' outbox counts are looked at only where k_exec has announced an arrival -- TSC_ARR_BIT in this
  //  drivable's own word)
  const bool announced = (hdr & TSC_ARR_BIT) != 0;
  int o_feed[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (announced && fb + q < fe) {
      const int f = is_link ? N.link_start_lane[d - N.n_lanes] : N.lane_in_links[fb + q];
      o_feed[q] = TSC_OB_OF(S.cnt[IDX(f)]);
    }
  const int n = hdr & TSC_CNT_MASK;
  const int m_self = n > 0 ? (o_raw < TSC_OB_MAX ? o_raw : TSC_OB_MAX) : 0;
  // ---- is anything arriving?
  int n_arr = 0, first_f = -1; // arrivals, and the outbox (drivable, index) of the first one found
  int first_k = 0;
  if (is_link) {
    const int f = N.link_start_lane[d - N.n_lanes];
    const int m = o_feed[0] < TSC_OB_MAX ? o_feed[0] : TSC_OB_MAX;
    for (int k = 0; k < m; k++) {
      size_t ox = ((size_t)f * TSC_OB_MAX + k) * (size_t)S.RS + (size_t)r;
      if (S.obx[ox].aux == d) {
        if (n_arr == 0) {
          first_f = f;
          first_k = k;
        }
        n_arr++;
      }
    }
  } else {
    // every vehicle leaving a feeder lane-link comes here
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (fb + q < fe && o_feed[q]) {
        if (n_arr == 0) first_f = N.lane_in_links[fb + q];
        n_arr += o_feed[q] < TSC_OB_MAX ? o_feed[q] : TSC_OB_MAX;
      }
    for (int q = fb + 4; announced && q < fe; q++) {
      int m = TSC_OB_OF(S.cnt[IDX(N.lane_in_links[q])]);
      if (m > TSC_OB_MAX) m = TSC_OB_MAX;
      if (m) {
        if (n_arr == 0) first_f = N.lane_in_links[q];
        n_arr += m;
      }
    }
  }
  const bool arrivals = n_arr > 0;
  bool spawn_possible = false;
  if (!is_link && N.lane_is_entry[d]) {
    int set = set_r;
    int qb = N.spawn_ptr[set * N.n_lanes + d], qe = N.spawn_ptr[set * N.n_lanes + d + 1];
    if (qb != qe) {
      int c = S.spawn_cur[IDX(d)];
      if (qb + c < qe && N.veh_start[N.set_veh_ptr[set] + N.spawn_vid[qb + c]] <= t_now + 1) spawn_possible = true;
    }
  }
  if (m_self == 0 && !arrivals && !spawn_possible) {
    // quiet drivable: only the tail record moves (and the inserted-this-step bit clears)
    // (its next tail record was written with the last vehicle's result)
    if (hdr & TSC_NEW_BIT) {
      S.tail_tag[IDX(d)] = S.tag[SIDX(sb + n - 1)];
      S.tail_prev[IDX(d)] = -1;
      S.cnt[IDX(d)] = n;
    }
    return;
  }
  const int vbase = N.set_veh_ptr[set_r];
  int tp = (hdr & TSC_NEW_BIT) ? -1 : S.tail_prev[IDX(d)];
  // ---- keep / retire
  const size_t oxs = (size_t)d * TSC_OB_MAX * (size_t)S.RS + (size_t)r;
  const int s0 = m_self > 0 ? S.obx[oxs].slot : -1;
  const int s1 = m_self > 1 ? S.obx[oxs + (size_t)S.RS].slot : -1;
  const int s2 = m_self > 2 ? S.obx[oxs + 2 * (size_t)S.RS].slot : -1;
  for (int k = 0; k < m_self; k++) {
    size_t ox = ((size_t)d * TSC_OB_MAX + k) * (size_t)S.RS + (size_t)r;
    if (!is_link && S.obx[ox].aux < 0) { // left the network
      int vid = TSC_TAG_VID(S.obx[ox].tag);
      int st0 = N.veh_start[vbase + vid];
      atomicAdd(&S.fin_cnt[r], 1);
      atomicAdd(&S.fin_dsum[r], (unsigned long long)(t_now - st0));
      atomicAdd(&S.fin_ssum[r], (unsigned long long)st0);
    }
  }
  if (!is_link && m_self > 0) S.fz[IDX(d)] = 0; // slots shift: k_enum's standing block is void
  int w = 0;
  bool have_tail = false, old_tail_kept = false;
  double tl_dist = 0, tl_speed = 0;
  int tl_tag = 0;
  if (m_self == 0) {
    w = n;
    if (n > 0) {
      have_tail = true;
      old_tail_kept = true;
      const double2 tv = q_ds[DIDX(sb + n - 1)];
      tl_dist = tv.x;
      tl_speed = tv.y;
      tl_tag = S.tag[SIDX(sb + n - 1)];
    }
  } else {
    // leavers are (almost always) the first k vehicles of the list: then the rest shifts
    // down by k, moved four slots per turn with all loads issued before the stores
    const int k = m_self;
    bool prefix = (s0 < k) && (m_self < 2 || s1 < k) && (m_self < 3 || s2 < k);
    if (prefix) {
      for (int i = k; i < n; i += 4) {
        double2 vv[4];
        int2 mm[4];
        int tg[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (i + u < n) {
            vv[u] = q_ds[DIDX(sb + i + u)];
            mm[u] = S.ab[SIDX(sb + i + u)];
            tg[u] = S.tag[SIDX(sb + i + u)];
          }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (i + u < n) {
            q_ds[DIDX(sb + i + u - k)] = vv[u];
            S.ab[SIDX(sb + i + u - k)] = mm[u];
            S.tag[SIDX(sb + i + u - k)] = tg[u];
            tl_dist = vv[u].x;
            tl_speed = vv[u].y;
            tl_tag = tg[u];
          }
      }
      w = n - k;
      have_tail = w > 0;
      old_tail_kept = w > 0;
    } else {
      for (int i = 0; i < n; i++) {
        if (i == s0 || i == s1 || i == s2) continue;
        const double2 vi = q_ds[DIDX(sb + i)];
        double nd = vi.x, ns = vi.y;
        int tg = S.tag[SIDX(sb + i)];
        if (w != i) {
          q_ds[DIDX(sb + w)] = vi;
          S.tag[SIDX(sb + w)] = tg;
          S.ab[SIDX(sb + w)] = S.ab[SIDX(sb + i)];
        }
        have_tail = true;
        tl_dist = nd;
        tl_speed = ns;
        tl_tag = tg;
        old_tail_kept = (i == n - 1);
        w++;
      }
    }
  }
  if (!have_tail) tp = -1;
  else if (!old_tail_kept) {
    // the previous tail left while others stayed (overlap corner case): recover
    // the new tail's previous drivable from its route
    int vid = TSC_TAG_VID(tl_tag), seq = TSC_TAG_SEQ(tl_tag);
    tp = seq > 0 ? N.route_drv[N.veh_rrange[vbase + vid].x + seq - 1] : -1;
  }
  // ---- pull newcomers from feeder outboxes, in descending buffered distance
  if (arrivals) {
    double last_nd = 1e300;
    int last_vid = -1;
    for (int pass = 0; pass < 16; pass++) {
      double best_nd = -1e300;
      int best_vid = 0x7fffffff, best_f = -1;
      size_t best_ox = 0;
      int tg = 0, nblk = 0;
      double ns = 0;
      bool have_rec = false;
      if (n_arr == 1) {
        // the usual case, one arrival: no ordering needed, the whole record in one round trip
        if (pass > 0) break;
        best_f = first_f;
        best_ox = ((size_t)first_f * TSC_OB_MAX + first_k) * (size_t)S.RS + (size_t)r;
        const ObxRec rec = S.obx[best_ox];
        best_nd = rec.dist;
        tg = rec.tag;
        ns = rec.speed;
        nblk = rec.blk;
        best_vid = TSC_TAG_VID(tg);
        have_rec = true;
      } else
      for (int q = fb; q < fe; q++) {
        int f = is_link ? N.link_start_lane[d - N.n_lanes] : N.lane_in_links[q];
        int m = TSC_OB_OF(S.cnt[IDX(f)]);
        if (m > TSC_OB_MAX) m = TSC_OB_MAX;
        for (int k = 0; k < m; k++) {
          size_t ox = ((size_t)f * TSC_OB_MAX + k) * (size_t)S.RS + (size_t)r;
          int ax = S.obx[ox].aux;
          if (is_link && ax != d) continue; // lanes: every leaver of a feeder link comes here
          double nd = S.obx[ox].dist;
          int vid = TSC_TAG_VID(S.obx[ox].tag);
          bool after_last = nd < last_nd || (nd == last_nd && vid > last_vid);
          bool before_best = nd > best_nd || (nd == best_nd && vid < best_vid);
          if (after_last && before_best) {
            best_nd = nd;
            best_vid = vid;
            best_f = f;
            best_ox = ox;
          }
        }
      }
      if (best_f < 0) break;
      last_nd = best_nd;
      last_vid = best_vid;
      if (w >= cap) {
        atomicAdd(&S.errors[0], 1);
        break;
      }
      if (!have_rec) {
        const ObxRec rec = S.obx[best_ox];
        tg = rec.tag;
        ns = rec.speed;
        nblk = rec.blk;
      }
      int seq = TSC_TAG_SEQ(tg) + 1;
      int ntag = TSC_MAKE_TAG(best_vid, seq);
      int naux;
      if (is_link) {
        naux = t_now; // enterLaneLinkTime
      } else {
        const int2 rr = N.veh_rrange[vbase + best_vid]; // the vehicle's slice of route_drv
        naux = (rr.x + seq + 1 < rr.y) ? N.route_drv[rr.x + seq + 1] : -1;
      }
      q_ds[DIDX(sb + w)] = make_double2(best_nd, ns);
      S.tag[SIDX(sb + w)] = ntag;
      S.ab[SIDX(sb + w)] = make_int2(naux, nblk);
      have_tail = true;
      tl_dist = best_nd;
      tl_speed = ns;
      tl_tag = ntag;
      tp = is_link ? -1 : best_f;
      w++;
    }
  }
  // ---- tail record (the tail BEFORE the next step's insertion)
  if (have_tail) {
    S.ntail[IDX(d)] = make_double2(tl_dist, tl_speed);
    S.tail_tag[IDX(d)] = tl_tag;
  }
  S.tail_prev[IDX(d)] = tp;
  // ---- spawn for the next step (engine: flows + handleWaiting, appendix B.3)
  int has_new = 0;
  if (spawn_possible) {
    int tail_tmpl = have_tail ? tmpl_of<SINGLE>(N, vbase, TSC_TAG_VID(tl_tag)) : 0;
    spawn_if_due<SINGLE>(N, S, q_ds, r, d, sb, t_now + 1, w, have_tail, tl_dist, tail_tmpl, has_new);
  }
  // (the outbox count stays in the word: the drivables its leavers go to read it in this very launch)
  S.cnt[IDX(d)] = w | (has_new ? TSC_NEW_BIT : 0) | (hdr & TSC_OB_BITS);
  // ---- the cross-candidate masks of the intersections (k_notify's three conditions), where this drivable's change has
  //      changed them: a lane-link's own "vehicles on it" bit when it became empty or stopped being so; a lane's feeder
  //      links' "the tail of my end lane came from me" bits when its tail is another vehicle than before (an arrival, a
  //      spawn, the old tail gone); a lane's out-links' "the head of my start lane is heading for me" bits when its head is
  //      (leavers; the first vehicle of an empty lane).  (Quiet drivables change nothing: the bit cleared with a fresh
  //      tail's NEW flag had tail_prev -1 behind it.)
  {
    const int nl_ = N.n_lanes;
    if (is_link) {
      if ((w > 0) != (n > 0)) {
        const int k = d - nl_;
        unsigned long long *m = &S.arec[IDX(N.link_agent[k])].ov;
        const unsigned long long bit = 1ull << N.link_local[k];
        if (w > 0) atomicOr(m, bit);
        else atomicAnd(m, ~bit);
      }
    } else {
      const bool tail_changed = arrivals || has_new || (m_self > 0 && !old_tail_kept) || ((hdr & TSC_NEW_BIT) != 0);
      const bool head_changed = m_self > 0 || (n == 0 && w > 0);
      if (__builtin_amdgcn_ballot_w64(tail_changed || head_changed)) {
        const LaneMaskRec lm = N.lane_mask[d]; // (one 64-byte scalar load: the lane's feeder links and out-links)
        if (tail_changed && lm.ag_in >= 0) {
          unsigned long long *m = &S.arec[IDX(lm.ag_in)].oe;
          atomicAnd(m, ~lm.fm);
          if (w > 0 && !has_new && tp >= nl_) {
            unsigned long long bit = 0ull;
            if (lm.fid[0] == -2) bit = 1ull << N.link_local[tp - nl_]; // (more than four feeder links)
            else {
#pragma unroll
              for (int q = 0; q < 4; q++)
                if (tp == lm.fid[q]) bit = 1ull << ((lm.floc >> (8 * q)) & 63u);
            }
            atomicOr(m, bit);
          }
        }
        if (head_changed && lm.ag_out >= 0) {
          unsigned long long *m = &S.arec[IDX(lm.ag_out)].os;
          atomicAnd(m, ~lm.om);
          if (w > 0) {
            const int ax = S.ab[SIDX(sb)].x; // (the head's next lane-link; -1: its route ends on this lane)
            unsigned long long bit = 0ull;
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (ax == lm.oid[q]) bit = 1ull << ((lm.oloc >> (8 * q)) & 63u);
            if (bit) atomicOr(m, bit);
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------- k_reset
template <bool SINGLE>
__global__ __launch_bounds__(256) void k_reset(DevNet N, DevState S, DevCfg C, const unsigned char *mask, int n_rb,
                                               int n_dblk) {
  int rb, dblk;
  if (!map_block(n_rb, n_dblk, rb, dblk)) return;
  int d = __builtin_amdgcn_readfirstlane(dblk * (int)blockDim.y + (int)threadIdx.y);
  int r = rb * 64 + threadIdx.x;
  if (d >= N.n_drv || r >= S.R) return;
  if (mask && !mask[r]) return;
  // S.set / S.t were prepared by k_reset_replica (previous launch)
  S.tail_prev[IDX(d)] = -1;
  int w = 0, has_new = 0;
  if (d < N.n_lanes) {
    S.spawn_cur[IDX(d)] = 0;
    S.fz[IDX(d)] = 0;
    spawn_if_due<SINGLE>(N, S, S.ds, r, d, N.drv_slot_base[d], 0, w, false, 0.0, 0, has_new);
  }
  S.cnt[IDX(d)] = w | (has_new ? TSC_NEW_BIT : 0);
}

__global__ void k_reset_replica(DevNet N, DevState S, DevCfg C, const unsigned char *mask, const int *sets, int first) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= S.R) return;
  if (mask && !mask[r]) return;
  if (S.parked) S.parked[r] = 0; // (a reset starts a parked replica again)
  int restarts = first ? 0 : S.restarts[r] + (C.sequential_sets ? 1 : 0);
  S.restarts[r] = restarts;
  int set = (C.set_offset + r + restarts) % N.n_sets;
  if (sets && sets[r] >= 0) set = sets[r] % N.n_sets; // the caller's draw (flow_rate_type: random)
  S.set[r] = set;
  S.t[r] = -1;
  S.fin_cnt[r] = 0;
  S.fin_dsum[r] = 0;
  S.fin_ssum[r] = 0;
  for (int a = 0; a < N.n_agents; a++) {
    S.cur_idx[IDX(a)] = 0;
    S.top[IDX(a)] = 0;
    S.changed[IDX(a)] = 0;
    S.arec[IDX(a)].raw = N.agent_phase_raw[a * TSC_MAX_AGENT_PHASES];
  }
}

// ------------------------------------------------------------------ k_phase
// action_space.apply -> switch_phase -> update_current_phase
// (reference common/actions.py:99-108,144-158; common/traffic_signal.py:94-109)
__global__ __launch_bounds__(256) void k_phase(DevNet N, DevState S, DevCfg C, const int *actions, int n_rb, int n_ablk) {
  int rb, ablk;
  if (!map_block(n_rb, n_ablk, rb, ablk)) return;
  int a = __builtin_amdgcn_readfirstlane(ablk * (int)blockDim.y + (int)threadIdx.y);
  int r = rb * 64 + threadIdx.x;
  if (a >= N.n_agents || r >= S.R || TSC_PARKED(S, r)) return;
  int act = actions[(size_t)r * N.n_agents + a];
  if (act == -1) return; // signal left untouched (no switch_phase call in the reference)
  int np_ = N.agent_n_phases[a];
  int cur = S.cur_idx[IDX(a)];
  int target;
  if (C.action_space == TSC_ACT_PHASE_SWITCH) target = act == 1 ? (cur + 1) % np_ : cur;
  else target = act;
  if (target < 0 || target >= np_) {
    atomicAdd(&S.errors[2], 1);
    target = cur;
  }
  if (target == cur) {
    S.changed[IDX(a)] = 0;
    S.top[IDX(a)] += C.yellow_time;
  } else {
    S.changed[IDX(a)] = 1;
    S.top[IDX(a)] = C.yellow_time;
  }
  S.cur_idx[IDX(a)] = target;
  S.arec[IDX(a)].raw = N.agent_phase_raw[a * TSC_MAX_AGENT_PHASES + target];
}

// ---------------------------------------------------------------- k_measure
// std::to_string(double) -> float(): value rounded to 6 decimals
__device__ __forceinline__ double round6(double x) {
  double k = rint(x * 1e6);
  double rr = fma(x, 1e6, -k);
  if (rr > 0.5) k += 1.0;
  else if (rr < -0.5) k -= 1.0;
  return k / 1e6;
}
// CPython float floor division a // b for a >= 0, b > 0.  CPython derives it from the exact
// fmod and repairs the rounding of (a - mod) / b, so the result is the mathematical
// floor(a / b); when the remainder of the rounded quotient is clear of both bin edges
// (by 1e-9 b, far above the rounding of one fma) that quotient's floor is the same
// number, and the fmod loop is only run for values sitting on an edge.
__device__ __forceinline__ double py_floordiv(double a, double b) {
  {
    const double k = floor(a / b);
    const double rem = fma(-k, b, a);
    const double tol = 1e-9 * b;
    if (a >= 0.0 && rem > tol && rem < b - tol) return k;
  }
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && (mod < 0.0)) div -= 1.0;
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

#define MEASURE_WAVES 4
#define MAX_VIS 16
// Lane measurements (reference retriever.py:54-85, :20-52): counts, queue, speed sum in
// list order, and the two visibility windows of the position matrix.  A bin of the
// matrix holds x = -1 + sum over its vehicles of (1 + speed / max speed), added in list
// order; what the observation needs of it are three predicates -- x >= 0, x == 0.0,
// x >= 1.0 -- and with every term non-negative those follow from the number of vehicles
// in the bin (0, 1, more) and, for a single vehicle, from its own term (x is then exactly
// that term: (-1 + 1) + s): kept as bit masks in registers, one bit per window bin.
__global__ __launch_bounds__(64 * MEASURE_WAVES) void k_measure(DevNet N, DevState S, DevCfg C, int n_rb, int n_lblk) {
  int rb, lblk;
  if (!map_block(n_rb, n_lblk, rb, lblk)) return;
  int l = __builtin_amdgcn_readfirstlane(lblk * (int)blockDim.y + (int)threadIdx.y);
  int r = rb * 64 + threadIdx.x;
  if (l >= N.n_lanes || r >= S.R) return;
  const int vis = C.visibility;
  int hdr = S.cnt[IDX(l)];
  int n = (hdr & TSC_CNT_MASK) - ((hdr & TSC_NEW_BIT) ? 1 : 0);
  const int sb = N.drv_slot_base[l];
  const int bins = N.lane_bins[l];
  const double len = N.lane_len_pytsc[l];
  const double maxspd = N.drv_max_speed[l];
  // padded matrix length and the two windows [-vis:] and [:vis]
  const int plen = bins < vis ? vis : bins;
  const int in0 = plen - vis;
  unsigned int in_one = 0, in_two = 0, in_zero = 0, in_ge1 = 0, out_one = 0;
  int q = 0;
  double sum = 0;
  double bin_size = bins > 0 ? len / bins : 1.0;
  // observation noise (round 5: on this family too): the value of every incoming-window bin, -1 + sum(1 + s / s_max) in
  // vehicle order (f32: the noise is compared with it, nothing else reads it); this thread owns the lane's row of them
  float *const val = S.lane_in_val ? S.lane_in_val + ((size_t)l * vis) * S.RS + r : nullptr;
  if (val)
    for (int b = 0; b < vis; b++) val[(size_t)b * S.RS] = -1.0f;
  const double inv_bin = 1.0 / bin_size;
  constexpr int MU = 4; // records requested together (8: +7 % time on the 16x16 grid -- 82 VGPRs, fewer wavefronts in flight)
  for (int i0 = 0; __builtin_amdgcn_ballot_w64(i0 < n); i0 += MU) {
    double2 v[MU];
#pragma unroll
    for (int u = 0; u < MU; u++)
      if (i0 + u < n) v[u] = S.ds[DIDX(sb + i0 + u)];
#pragma unroll
    for (int u = 0; u < MU; u++) {
      if (i0 + u >= n) continue;
      const double dist = v[u].x, speed = v[u].y;
      if (speed < 0.1) q++;
      sum += speed;
      if (bins <= 0) continue;
      // (round 4, as the resident family's measure phase does: a vehicle clear of both windows -- by a millimetre, a
      //  thousand times what round6 and the floor division can move it -- sets no window bit: most of a lane's length;
      //  and floor(round(dist, 6) // bin size) as CPython computes it is the mathematical floor, which a guess from the
      //  reciprocal IS whenever the vehicle is clear of the guessed bin's edges by 1e-5 and of the lane's end; the
      //  divisions -- two and an fmod per vehicle, a third of this kernel's cycles were instruction issue -- are for
      //  the others.  tests/test_measure_fastpath.py holds the host-side proof of both.)
      if (dist > vis * bin_size + 1e-3 && dist < in0 * bin_size - 1e-3) continue;
      int idx;
      {
        const double kq = floor(dist * inv_bin);
        const double rem = fma(-kq, bin_size, dist);
        if (kq >= 0.0 && rem > 1e-5 && rem < bin_size - 1e-5 && dist < len - 1e-5) {
          idx = (int)kq;
        } else {
          double pos = round6(dist);
          if (pos < 0) pos = 0;
          else if (pos > len) pos = len;
          idx = (int)py_floordiv(pos, bin_size);
        }
      }
      if (idx >= bins) idx = bins - 1;
      if (idx < vis) out_one |= 1u << idx;
      if (idx >= in0) {
        const unsigned int bit = 1u << (idx - in0);
        if (val) val[(size_t)(idx - in0) * S.RS] += (float)(1.0 + (speed == 0.0 ? 0.0 : round6(speed) / maxspd));
        if (in_one & bit) {
          in_two |= bit;
        } else {
          in_one |= bit;
          const double norm_speed = speed == 0.0 ? 0.0 : round6(speed) / maxspd;
          if (norm_speed == 0.0) in_zero |= bit;
          if (norm_speed >= 1.0) in_ge1 |= bit;
        }
      }
    }
  }
  // 3 predicate bits per incoming-window bin: occupied (>= 0), zero (== 0), >= 1
  unsigned long long ib = 0; // 3 bits x MAX_VIS (16) bins = 48 bits
  for (int b = 0; b < vis; b++) {
    const unsigned int one = (in_one >> b) & 1u, two = (in_two >> b) & 1u;
    if (one) ib |= 1ull << (3 * b);
    if (one && !two && ((in_zero >> b) & 1u)) ib |= 2ull << (3 * b);
    if (two || (one && ((in_ge1 >> b) & 1u))) ib |= 4ull << (3 * b);
  }
  S.lane_n[IDX(l)] = n;
  S.lane_q[IDX(l)] = q;
  if (q) atomicAdd(&S.nq_all[r], q);
  S.lane_sum[IDX(l)] = sum;
  S.lane_in_bits[IDX(l)] = ib;
  S.lane_out_bits[IDX(l)] = out_one;
}

// Full position matrices for the single-replica facade / parity checks.
__global__ void k_posmat(DevNet N, DevState S, DevCfg C, double *posmat, int max_bins, int n_rb, int n_lblk) {
  int rb, lblk;
  if (!map_block(n_rb, n_lblk, rb, lblk)) return;
  int l = lblk * (int)blockDim.y + (int)threadIdx.y;
  int r = rb * 64 + threadIdx.x;
  if (l >= N.n_lanes || r >= S.R) return;
  int hdr = S.cnt[IDX(l)];
  int n = (hdr & TSC_CNT_MASK) - ((hdr & TSC_NEW_BIT) ? 1 : 0);
  const int sb = N.drv_slot_base[l];
  const int bins = N.lane_bins[l];
  const double len = N.lane_len_pytsc[l];
  const double maxspd = N.drv_max_speed[l];
  double *pm = posmat + ((size_t)r * N.n_lanes + l) * max_bins;
  for (int b = 0; b < max_bins; b++) pm[b] = -1.0;
  if (bins <= 0) return;
  double bin_size = len / bins;
  for (int i = 0; i < n; i++) {
    const double2 vi = S.ds[DIDX(sb + i)];
    double pos = round6(vi.x);
    if (pos < 0) pos = 0;
    else if (pos > len) pos = len;
    int idx = (int)py_floordiv(pos, bin_size);
    if (idx >= bins) idx = bins - 1;
    double norm_speed = round6(vi.y) / maxspd;
    pm[idx] += 1.0;
    pm[idx] += norm_speed;
  }
}

__global__ void k_lane_detail(DevNet N, DevState S, int *lane_counts, double *lane_speed, int n_rb, int n_lblk) {
  int rb, lblk;
  if (!map_block(n_rb, n_lblk, rb, lblk)) return;
  int l = lblk * (int)blockDim.y + (int)threadIdx.y;
  int r = rb * 64 + threadIdx.x;
  if (l >= N.n_lanes || r >= S.R) return;
  lane_counts[((size_t)r * N.n_lanes + l) * 2 + 0] = S.lane_n[IDX(l)];
  lane_counts[((size_t)r * N.n_lanes + l) * 2 + 1] = S.lane_q[IDX(l)];
  lane_speed[(size_t)r * N.n_lanes + l] = S.lane_sum[IDX(l)];
}

__global__ void k_lane_window_counts(DevNet N, DevState S, DevCfg C, int *out, int n_rb, int n_lblk) {
  int rb, lblk;
  if (!map_block(n_rb, n_lblk, rb, lblk)) return;
  int l = lblk * (int)blockDim.y + (int)threadIdx.y;
  int r = rb * 64 + threadIdx.x;
  if (l >= N.n_lanes || r >= S.R) return;
  const unsigned long long ib = S.lane_in_bits[IDX(l)];
  const unsigned int ob = S.lane_out_bits[IDX(l)];
  int nz = 0, no = 0;
  for (int b = 0; b < C.visibility; b++) {
    unsigned int t3 = (unsigned int)(ib >> (3 * b)) & 7u;
    nz += (t3 >> 1) & 1u;
    no += t3 & 1u;
  }
  int *o = out + ((size_t)r * N.n_lanes + l) * 3;
  o[0] = nz;
  o[1] = no;
  o[2] = __builtin_popcount(ob);
}

// ------------------------------------------------------------------ k_agent
__device__ __forceinline__ double lane_mean_speed(const DevState &S, int r, int l) {
  int n = S.lane_n[IDX(l)];
  return n == 0 ? 0.0 : S.lane_sum[IDX(l)] / n;
}
__device__ __forceinline__ double lane_occupancy(const DevNet &N, const DevState &S, const DevCfg &C, int r, int l) {
  // (lane_slots = lane_len_pytsc / veh_size_min_gap, the host's IEEE division: the same bits as dividing here, once per
  //  lane and replica less)
  return S.lane_n[IDX(l)] / N.lane_slots[l];
}

#define AGENT_WAVES 2
// (round 4: plain stores, not non-temporal ones.  A row is 1 296 bytes: the 128-byte lines at its two ends are shared with
//  the neighbouring signals' rows, which other wavefronts write within microseconds; through the L2 the halves merge,
//  streamed past it they are partial-line writes: k_agent 0.42-0.46 -> 0.38 ms on the 16x16 grid, same box)
#ifdef TSC_AGENT_NT_STORE
#define TSC_AGENT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define TSC_AGENT_STORE(v, p) (*(p) = (v))
#endif
// The 64 rows of one wavefront (replicas rb * 64 ..., signal a; bytes, value + 1, in LDS) -> rows of floats of [R, A, size].
// Round 4: every lane stores FOUR floats per instruction (16 bytes, non-temporal: written once, not read by this library) and
// the (row, position) pairs of the 64 rows are dealt to the lanes as one index space -- 64 x 81 float4 of the observation
// rows are 81 store instructions per wavefront; a float per lane and one row at a time took 384 (the launch was bound by
// store issue: 0.83 ms per env step on the 16x16 grid for 1.36 GB).  Same bytes in memory.
__device__ __forceinline__ void agent_store_rows(const unsigned char *rows, int row_stride, float *out, int size, int rb, int a, int A,
                                                 int R, int lx) {
  const int n_rows = min(64, R - rb * 64);
  if ((size & 3) == 0) {
    typedef float a_f4 __attribute__((ext_vector_type(4)));
    const int q = size >> 2; // float4 per row
    int rr = lx / q, j4 = lx - rr * q;
    while (rr < n_rows) {
      const unsigned int w4 = *(const unsigned int *)(rows + (size_t)rr * row_stride + 4 * j4);
      a_f4 v;
      // (byte k to float in one operation -- v_cvt_f32_ubyte<k> -- and the offset as a float add: two operations per value)
      v.x = (float)(w4 & 255u) - 1.0f;
      v.y = (float)((w4 >> 8) & 255u) - 1.0f;
      v.z = (float)((w4 >> 16) & 255u) - 1.0f;
      v.w = (float)(w4 >> 24) - 1.0f;
      TSC_AGENT_STORE(v, (a_f4 *)(out + ((size_t)(rb * 64 + rr) * A + a) * size) + j4);
      j4 += 64;
      while (j4 >= q) {
        j4 -= q;
        rr++;
      }
    }
  } else {
    for (int rr = 0; rr < n_rows; rr++) {
      float *dst = out + ((size_t)(rb * 64 + rr) * A + a) * size;
      const unsigned char *src = rows + (size_t)rr * row_stride;
      for (int j = lx; j < size; j += 64) dst[j] = (float)((int)src[j] - 1);
    }
  }
}

// The same for observation rows held as 2-bit symbols (the value -1, 0 or 1 in two's complement: a signed bit-field extract
// and a conversion per value; `ow` words per row): a float4 is one byte.
__device__ __forceinline__ void agent_store_rows2(const unsigned int *rows, int ow, float *out, int size, int rb, int a, int A, int R,
                                                  int lx) {
  typedef float a_f4 __attribute__((ext_vector_type(4)));
  const int n_rows = min(64, R - rb * 64);
  const int q = size >> 2; // float4 per row
  int rr = lx / q, j4 = lx - rr * q;
  while (rr < n_rows) {
    const unsigned int b = (rows[(size_t)rr * ow + (j4 >> 2)] >> (8 * (j4 & 3))) & 255u;
    a_f4 v;
    v.x = (float)((int)(b << 30) >> 30);
    v.y = (float)((int)(b << 28) >> 30);
    v.z = (float)((int)(b << 26) >> 30);
    v.w = (float)((int)(b << 24) >> 30);
    TSC_AGENT_STORE(v, (a_f4 *)(out + ((size_t)(rb * 64 + rr) * A + a) * size) + j4);
    j4 += 64;
    while (j4 >= q) {
      j4 -= q;
      rr++;
    }
  }
}

// Per-agent statistics, observation / state vectors, action mask (reference
// backends/cityflow/traffic_signal.py:101-141, common/observations.py:140-160,
// 183-205, common/actions.py:119-131,169-188, common/traffic_signal.py:329-404).
// Rows are assembled in LDS (one row per replica of the wave) and streamed out
// so that the [R, A, obs] stores are contiguous 256-byte segments.
__global__ __launch_bounds__(64 * AGENT_WAVES) void k_agent(DevNet N, DevState S, DevCfg C, tsc_step_out_t out,
                                                            int row_stride, int obs_words, int n_rb, int n_ablk) {
  // state rows: [AGENT_WAVES][64][row_stride] bytes, values stored + 1; observation rows: [AGENT_WAVES][64][obs_words]
  // words of 2-bit symbols (obs_words > 0; 0: bytes like the state rows)
  extern __shared__ unsigned char rows_lds[];
  int rb, ablk;
  bool live = map_block(n_rb, n_ablk, rb, ablk);
  int a = __builtin_amdgcn_readfirstlane(ablk * (int)blockDim.y + (int)threadIdx.y);
  int r = rb * 64 + threadIdx.x;
  const int wy = threadIdx.y, lx = threadIdx.x;
  live = live && a < N.n_agents;
  const bool mine = live && r < S.R && !TSC_PARKED(S, r);
  const int A = N.n_agents, vis = C.visibility;
  int cur = 0, np_ = 1;
  // Round 4: the per-lane words of the signal's incoming and outgoing lanes are REQUESTED together, before the first one
  // is used (clamped indices, no branches: the loads of a straight-line block travel together).  The loops "load a lane's
  // words, add them up" were some sixty dependent round trips per thread, with LDS rows limiting the CU to six wavefronts
  // to hide them behind: ~70 us per workgroup round, 0.74 ms per env step on the 16x16 grid for a launch that moves
  // 1.6 GB (profiles/r04_c16_kernel_stats.csv).  The sums below add the same terms in the same (lane) order.
  constexpr int MI = TSC_MAX_CTRL_LANES;
  int lq[MI], ln[MI], lon[MI];
  double lsum[MI];
  unsigned long long lbits[MI];
  const int ib = live ? N.agent_inc_ptr[a] : 0, n_inc = live ? N.agent_inc_ptr[a + 1] - ib : 0;
  const int ob = live ? N.agent_out_ptr[a] : 0, n_out = live ? N.agent_out_ptr[a + 1] - ob : 0;
  if (mine) {
#pragma unroll
    for (int k = 0; k < MI; k++) {
      const int l = N.agent_inc_lanes[ib + (k < n_inc ? k : n_inc - 1)];
      lq[k] = S.lane_q[IDX(l)];
      ln[k] = S.lane_n[IDX(l)];
      lsum[k] = S.lane_sum[IDX(l)];
      lbits[k] = (out.obs_dev || out.phase_scores_dev) ? S.lane_in_bits[IDX(l)] : 0ull;
      lon[k] = S.lane_n[IDX(N.agent_out_lanes[ob + (k < n_out ? k : n_out - 1)])];
    }
  }
  // ---- per-phase scores of the rule-based controllers (tsc_step_out_t::phase_scores_dev; reference
  //      controllers/controllers.py:93-112,153-178,227-242).  The movements of a phase are wave-uniform (one signal per
  //      wavefront): scalar loads, a loop over the set bits of the phase's lane mask; the per-lane counts travel packed
  //      eight bits each in 64-bit registers (no indexed register arrays); the row of 60 shorts leaves as 15 8-byte stores.
  if (out.phase_scores_dev && mine && N.agent_moves) {
    unsigned long long c_in[2] = {0ull, 0ull}, c_z[2] = {0ull, 0ull}, c_out[2] = {0ull, 0ull};
#pragma unroll
    for (int k = 0; k < MI; k++) {
      if (k < n_inc) {
        const unsigned long long b_ = lbits[k];
        c_in[k >> 3] |= (unsigned long long)__builtin_popcountll(b_ & 0x1249249249249249ull) << (8 * (k & 7));
        c_z[k >> 3] |= (unsigned long long)__builtin_popcountll((b_ >> 1) & 0x1249249249249249ull) << (8 * (k & 7));
      }
      if (k < n_out)
        c_out[k >> 3] |= (unsigned long long)__builtin_popcount(S.lane_out_bits[IDX(N.agent_out_lanes[ob + k])]) << (8 * (k & 7));
    }
    const int np_s = N.agent_n_phases[a];
    unsigned long long *dst = (unsigned long long *)(out.phase_scores_dev + ((size_t)r * A + a) * (TSC_MAX_PHASES * 3));
    unsigned long long acc = 0ull;
    int fill = 0; // bits of `acc` in use
    for (int p_ = 0; p_ < TSC_MAX_PHASES; p_++) {
      unsigned long long w3 = 0ull; // the phase's three scores, 16 bits each
      if (p_ < np_s) {
        const uint4 mv = N.agent_moves[a * TSC_MAX_AGENT_PHASES + p_];
        const unsigned long long sel = (unsigned long long)mv.y | ((unsigned long long)mv.z << 32);
        int s_q = 0, s_f = 0, s_p = 0;
        for (unsigned int m_ = mv.x & 0xFFFFu; m_; m_ &= m_ - 1u) {
          const int i = __builtin_ctz(m_), j = (int)((sel >> (4 * i)) & 15ull);
          const int n_in = (int)((c_in[i >> 3] >> (8 * (i & 7))) & 255ull), n_z = (int)((c_z[i >> 3] >> (8 * (i & 7))) & 255ull);
          const int n_o = (int)((c_out[j >> 3] >> (8 * (j & 7))) & 255ull);
          s_q += n_z;
          s_f += n_in;
          s_p += n_in > n_o ? n_in - n_o : n_o - n_in;
        }
        w3 = (unsigned long long)(unsigned short)s_q | ((unsigned long long)(unsigned short)s_f << 16) | ((unsigned long long)(unsigned short)s_p << 32);
      }
      acc |= w3 << fill;
      fill += 48;
      if (fill >= 64) {
        *dst++ = acc;
        fill -= 64;
        acc = fill ? (w3 >> (48 - fill)) : 0ull;
      }
    }
  }
  if (mine) {
    int nq = 0;
    double occ = 0, ms = 0, md = 0;
#pragma unroll
    for (int k = 0; k < MI; k++) {
      if (k < n_inc) {
        const int l = N.agent_inc_lanes[ib + k];
        nq += lq[k];
        occ += ln[k] / N.lane_slots[l];
        const double m = ln[k] == 0 ? 0.0 : lsum[k] / ln[k];
        ms += m;
        md += 1 - m / N.drv_max_speed[l];
      }
    }
    occ /= n_inc;
    ms /= n_inc;
    md /= n_inc;
    double oocc = 0;
#pragma unroll
    for (int k = 0; k < MI; k++)
      if (k < n_out) oocc += lon[k] / N.lane_slots[N.agent_out_lanes[ob + k]];
    for (int k = MI; k < n_out; k++) oocc += lane_occupancy(N, S, C, r, N.agent_out_lanes[ob + k]); // (more than 16 outgoing lanes)
    oocc /= n_out;
    double pressure = fabs(occ - oocc);
    cur = S.cur_idx[IDX(a)];
    np_ = N.agent_n_phases[a];
    int top = S.top[IDX(a)];
    int changed = S.changed[IDX(a)];
    double pc = changed ? 1.0 : 0.0;
    double local;
    if (C.reward_kind == 0) local = -C.flickering_coef * pc - nq - 1e-6;
    else local = -C.flickering_coef * pc - pressure - 1e-6;
    S.agent_local[IDX(a)] = local;
    S.agent_pressure[IDX(a)] = pressure;
    if (out.agent_stats_dev) {
      double *s = out.agent_stats_dev + ((size_t)r * A + a) * TSC_N_AGENT_STATS;
      int sim_step = (S.t[r] + 1) - C.initial_wait_time;
      s[TSC_A_N_QUEUED] = nq;
      s[TSC_A_OCCUPANCY] = occ;
      s[TSC_A_MEAN_SPEED] = ms;
      s[TSC_A_MEAN_DELAY] = md;
      s[TSC_A_OUT_OCCUPANCY] = oocc;
      s[TSC_A_PRESSURE] = pressure;
      s[TSC_A_NORM_TIME_ON_PHASE] = (double)top / (double)N.agent_phase_max[a * TSC_MAX_AGENT_PHASES + cur];
      s[TSC_A_SIM_STEP_NORM] = sim_step / 3600.0;
    }
    if (out.phase_dev) {
      int *p = out.phase_dev + ((size_t)r * A + a) * 3;
      p[0] = cur;
      p[1] = top;
      p[2] = changed;
    }
    if (out.mask_dev) {
      const int *green = N.agent_phase_green + a * TSC_MAX_AGENT_PHASES;
      int nxt = (cur + 1) % np_;
      unsigned int m = 0;
      if (green[cur]) {
        int mn = N.agent_phase_min[a * TSC_MAX_AGENT_PHASES + cur];
        int mx = N.agent_phase_max[a * TSC_MAX_AGENT_PHASES + cur];
        if (top < mn) m = 1u << cur;
        else if (top < mx) m = (1u << cur) | (1u << nxt);
        else if (top == mx) m = 1u << nxt;
      } else if (C.round_robin) {
        m = 1u << nxt;
      } else {
        for (int i = 0; i < np_; i++)
          if (green[i] && i != cur - 1) m |= 1u << i;
      }
      unsigned char *mk = out.mask_dev + ((size_t)r * A + a) * C.n_actions;
      if (C.action_space == TSC_ACT_PHASE_SWITCH) {
        mk[0] = (m >> cur) & 1u;
        mk[1] = (m >> nxt) & 1u;
      } else {
        for (int i = 0; i < C.n_actions; i++) mk[i] = (i < np_) ? ((m >> i) & 1u) : 0;
      }
    }
  }
  // ---- observation rows (Q1 integer truncation, Q2 compaction)
  // (round 4: a row's padding -- some 190 of an observation row's 324 values -- is not written value by value by the row's
  //  thread: the wavefront clears its 64 rows with 16-byte LDS stores first; and a lane's kept bins are walked by their
  //  bits, not by testing all `visibility` bins)
  const bool noisy = C.obs_noise_std > 0.0f && S.lane_in_val != nullptr && out.obs_dev != nullptr;
  const unsigned long long nkey0 = (noisy && mine) ? tsc_noise_key0(C.obs_noise_seed, C.replica_base + r, S.t[r] + 1) : 0ull;
  auto clear_rows = [&]() {
    uint4 *z = (uint4 *)(rows_lds + (size_t)wy * 64 * row_stride); // (64 x row_stride bytes: a multiple of 256)
    for (int i = lx; i < 4 * row_stride; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  };
  if (out.obs_dev && obs_words > 0) {
    // Round 4: an observation value is -1, 0 or 1 (Q1's truncation of values clipped to [0, 1] / [-1, 1]): a row is held as
    // 2-bit symbols (two's complement: 11, 00, 01), 84 bytes instead of 329 -- the rows were what limited a CU to six wavefronts of this kernel -- and a
    // thread writes its row as a bit stream through a register (21 word stores instead of ~135 byte stores and the
    // wavefront's clearing pass): the nine static features of a lane are one 18-bit constant (N.lane_static_pk), a kept bin
    // two bits; 16 lanes x 2 bits = one word per column of the body, so the phase one-hot starts on a word.
    unsigned int *rows2 = (unsigned int *)rows_lds;
    if (mine) {
      unsigned int *row = rows2 + ((size_t)wy * 64 + lx) * obs_words;
      unsigned long long acc = 0ull;
      int pos = 0, wi = 0;
      unsigned long long bin0 = 0ull; // bit 3b of `sel`: bin b is listed (occupied and not exactly zero: value > 0)
      for (int b = 0; b < vis; b++) bin0 |= 1ull << (3 * b);
#pragma unroll
      for (int k = 0; k < MI; k++) {
        if (k < n_inc) { // (no break: the loop must unroll, lbits are registers)
          acc |= (unsigned long long)N.lane_static_pk[N.agent_inc_lanes[ib + k]] << pos;
          pos += 2 * TSC_LANE_STATIC;
          if (pos >= 32) {
            row[wi++] = (unsigned int)acc;
            acc >>= 32;
            pos -= 32;
          }
          const unsigned long long bits = lbits[k];
          unsigned long long sel = bits & ~(bits >> 1) & bin0;
          unsigned int ord = 0;
          while (sel) { // kept iff value > 0; value = trunc(clip(v, 0, 1)); in bin order
            const int b3 = __builtin_ctzll(sel);
            sel &= sel - 1ull;
            unsigned long long one = (bits >> (b3 + 2)) & 1ull;
            if (noisy) { // _add_gaussian_noise (observations.py:72-88): one draw per surviving entry, clip, truncate
              const float v_ = S.lane_in_val[((size_t)N.agent_inc_lanes[ib + k] * vis + b3 / 3) * S.RS + r];
              one = (v_ + C.obs_noise_std * r_noise_normal(nkey0, (unsigned int)a, (unsigned int)k, ord++) >= 1.0f) ? 1ull : 0ull;
            }
            acc |= one << pos; // (value 1: symbol 01; value 0: 00)
            pos += 2;
            if (pos >= 32) {
              row[wi++] = (unsigned int)acc;
              acc >>= 32;
              pos -= 32;
            }
          }
        }
      }
      const int body_w = (C.obs_size - TSC_MAX_PHASES) >> 4; // (16 symbols per word; the body is 16 x (9 + visibility) long)
      if (wi < body_w) row[wi++] = (unsigned int)(acc | (~0ull << pos)); // the last word with values; the padding (-1: 11) behind it
      while (wi < body_w) row[wi++] = 0xFFFFFFFFu;
      // phase one-hot: 0.0 (00) for every phase the signal has, 1.0 (01) for the current one, -1.0 (11) beyond (np_ <= 20)
      const unsigned long long ph = (~0ull << (2 * np_)) | (1ull << (2 * cur));
      row[body_w] = (unsigned int)ph;
      row[body_w + 1] = (unsigned int)(ph >> 32);
    }
    __syncthreads();
    if (live) agent_store_rows2(rows2 + (size_t)wy * 64 * obs_words, obs_words, out.obs_dev, C.obs_size, rb, a, A, S.R, lx);
    __syncthreads();
  } else if (out.obs_dev) {
    clear_rows();
    if (mine) {
      unsigned char *row = rows_lds + ((size_t)wy * 64 + lx) * row_stride;
      int w = 0;
      // bit 3b of `sel`: bin b is listed (occupied and not exactly zero: value > 0)
      unsigned long long bin0 = 0ull;
      for (int b = 0; b < vis; b++) bin0 |= 1ull << (3 * b);
#pragma unroll
      for (int k = 0; k < MI; k++) {
        if (k < n_inc) { // (no break: the loop must unroll, lbits are registers)
          const int l = N.agent_inc_lanes[ib + k];
          for (int j = 0; j < TSC_LANE_STATIC; j++) row[w++] = (unsigned char)(N.lane_static[l * TSC_LANE_STATIC + j] + 1);
          const unsigned long long bits = lbits[k];
          unsigned long long sel = bits & ~(bits >> 1) & bin0;
          unsigned int ord = 0;
          while (sel) { // kept iff value > 0; value = trunc(clip(v, 0, 1)); in bin order
            const int b3 = __builtin_ctzll(sel);
            sel &= sel - 1ull;
            bool one = ((bits >> b3) & 4ull) != 0;
            if (noisy) {
              const float v_ = S.lane_in_val[((size_t)l * vis + b3 / 3) * S.RS + r];
              one = v_ + C.obs_noise_std * r_noise_normal(nkey0, (unsigned int)a, (unsigned int)k, ord++) >= 1.0f;
            }
            row[w++] = one ? 2 : 1;
          }
        }
      }
      const int body = C.obs_size - TSC_MAX_PHASES;
      for (int i = 0; i < TSC_MAX_PHASES; i++) row[body + i] = i < np_ ? (i == cur ? 2 : 1) : 0;
    }
    __syncthreads();
    if (live) agent_store_rows(rows_lds + (size_t)wy * 64 * row_stride, row_stride, out.obs_dev, C.obs_size, rb, a, A, S.R, lx);
    __syncthreads();
  }
  if (out.state_dev) {
    clear_rows();
    if (mine) {
      unsigned char *row = rows_lds + ((size_t)wy * 64 + lx) * row_stride;
      int w = 0;
#pragma unroll
      for (int k = 0; k < MI; k++) {
        if (k < n_inc) { // (no break: the loop must unroll, lq / ln / lsum are registers)
          const int l = N.agent_inc_lanes[ib + k];
          for (int j = 0; j < TSC_LANE_STATIC; j++) row[w++] = (unsigned char)(N.lane_static[l * TSC_LANE_STATIC + j] + 1);
          row[w++] = (unsigned char)(lq[k] + 1);
          row[w++] = (unsigned char)((int)(long long)(ln[k] / N.lane_slots[l]) + 1);
          row[w++] = (unsigned char)((int)(long long)(ln[k] == 0 ? 0.0 : lsum[k] / ln[k]) + 1);
        }
      }
      const int body = C.state_size - TSC_MAX_PHASES;
      for (int i = 0; i < TSC_MAX_PHASES; i++) row[body + i] = (i < np_ && i == cur) ? 2 : 1;
    }
    __syncthreads();
    if (live) agent_store_rows(rows_lds + (size_t)wy * 64 * row_stride, row_stride, out.state_dev, C.state_size, rb, a, A, S.R, lx);
  }
}

// ------------------------------------------------------------------ k_stats
__device__ double np_pairwise_sum_col(const double *a, size_t stride, int n) {
  // numpy add.reduce on a contiguous f64 array (n <= 128 path and recursion)
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; i++) res += a[(size_t)i * stride];
    return res;
  } else if (n <= 128) {
    double r0 = a[0], r1 = a[stride], r2 = a[2 * stride], r3 = a[3 * stride], r4 = a[4 * stride],
           r5 = a[5 * stride], r6 = a[6 * stride], r7 = a[7 * stride];
    int i;
#pragma unroll 4 /* (32 loads in flight: one thread per replica walks the column alone) */
    for (i = 8; i < n - (n % 8); i += 8) {
      r0 += a[(size_t)(i + 0) * stride];
      r1 += a[(size_t)(i + 1) * stride];
      r2 += a[(size_t)(i + 2) * stride];
      r3 += a[(size_t)(i + 3) * stride];
      r4 += a[(size_t)(i + 4) * stride];
      r5 += a[(size_t)(i + 5) * stride];
      r6 += a[(size_t)(i + 6) * stride];
      r7 += a[(size_t)(i + 7) * stride];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[(size_t)i * stride];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_col(a, stride, n2) + np_pairwise_sum_col(a + (size_t)n2 * stride, stride, n - n2);
  }
}

// The three ordered f64 sums over ALL lanes of the info dict (the reference uses Python's `sum`: sequential adds in lane
// order, so the order is part of the result) and the two integer totals, for 64 replicas per workgroup.  One thread per
// replica walking 3 264 lanes (16x16) is a chain of dependent memory round trips: 3.2 ms per env step, a quarter of the
// GPU time of that configuration (profiles/r03_c16_kernel_stats.csv).  Here the TERMS of a stage of NS_LANES lanes are
// computed by fifteen loader wavefronts (lane-of-wave <-> replica: coalesced 256 / 512-byte rows, the f64 divisions
// of different lanes side by side) into LDS, double-buffered, while wavefront 0 adds the previous stage's terms in lane
// order: the same operations on the same operands in the same order per replica, the round trips overlapped.
#define NS_WAVES 16
#define NS_LANES 30 /* lanes per stage: two per loader wavefront */
// (round 4: RC -- two of a term's three divisions, by the lane's slot count and by its speed limit, through div_c with the
//  host's reciprocals: fifteen loader wavefronts on ONE CU made eight f64 divisions per stage each, and that, not memory, was
//  what a stage lasted: 0.220 -> 0.174 ms on the 16x16 grid.  The third, speed sum / vehicle count, stays a division: its
//  reciprocal would be a gather from a table, and with that on the chain the gain is gone -- measured.)
template <bool RC>
__global__ __launch_bounds__(64 * NS_WAVES) void k_net_sums(DevNet N, DevState S, DevCfg C) {
  __shared__ double t_occ[2][NS_LANES][64], t_nm[2][NS_LANES][64], t_mv[2][NS_LANES][64];
  __shared__ int t_cnt[NS_WAVES][2][64];
  const int lx = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6);
  const int r = (int)blockIdx.x * 64 + lx;
  const bool rv = r < S.R;
  const int nl = N.n_lanes, n_stage = (nl + NS_LANES - 1) / NS_LANES;
  int nq = 0, nveh = 0;
  double dens = 0, nms = 0, tvs = 0;
  // (round 4: a stage's words are REQUESTED two stages ahead and turned into terms one stage ahead: with the request and
  //  the terms in one step a stage lasted as long as a round trip to memory -- 2.9 us x 109 stages on the 16x16 grid --
  //  with fifteen wavefronts waiting for six loads each)
  int rq_n[2] = {0, 0}, rq_q[2] = {0, 0};
  double rq_s[2] = {0, 0};
  auto request_stage = [&](int st) {
    if (w == 0) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int l = st * NS_LANES + (w - 1) * 2 + j;
      rq_n[j] = rq_q[j] = 0;
      rq_s[j] = 0.0;
      if (rv && l < nl) {
        rq_n[j] = S.lane_n[IDX(l)];
        rq_q[j] = S.lane_q[IDX(l)];
        rq_s[j] = S.lane_sum[IDX(l)];
      }
    }
  };
  auto terms_stage = [&](int st) { // (of the words request_stage(st) left in the registers)
    if (w == 0) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int slot = (w - 1) * 2 + j, l = st * NS_LANES + slot;
      double occ = 0.0, nm = 0.0, mv = 0.0;
      if (rv && l < nl) {
        const int n = rq_n[j];
        nq += rq_q[j];
        nveh += n;
        occ = div_r<RC>((double)n, N.lane_slots[l], N.lane_rslots[l]); // (lane_slots = lane_len_pytsc / veh_size_min_gap: the host's IEEE division)
        const double m = n == 0 ? 0.0 : rq_s[j] / n;
        nm = div_r<RC>(m, N.drv_max_speed[l], N.lane_rmax[l]);
        mv = m * n;
      }
      t_occ[st & 1][slot][lx] = occ;
      t_nm[st & 1][slot][lx] = nm;
      t_mv[st & 1][slot][lx] = mv;
    }
  };
  request_stage(0);
  terms_stage(0);
  if (n_stage > 1) request_stage(1);
  __syncthreads();
  for (int st = 0; st < n_stage; st++) {
    if (st + 1 < n_stage) terms_stage(st + 1);
    if (st + 2 < n_stage) request_stage(st + 2);
    if (w == 0) {
      const int cnt = nl - st * NS_LANES < NS_LANES ? nl - st * NS_LANES : NS_LANES;
      for (int j = 0; j < cnt; j++) { // (a lane beyond the last adds nothing: not even + 0.0)
        dens += t_occ[st & 1][j][lx];
        nms += t_nm[st & 1][j][lx];
        tvs += t_mv[st & 1][j][lx];
      }
    }
    __syncthreads();
  }
  t_cnt[w][0][lx] = nq;
  t_cnt[w][1][lx] = nveh;
  __syncthreads();
  if (w == 0 && rv) {
    int q = 0, v = 0;
    for (int i = 0; i < NS_WAVES; i++) {
      q += t_cnt[i][0][lx];
      v += t_cnt[i][1][lx];
    }
    S.net_sums[IDX(0)] = dens;
    S.net_sums[IDX(1)] = nms;
    S.net_sums[IDX(2)] = tvs;
    S.net_cnt[IDX(0)] = q;
    S.net_cnt[IDX(1)] = v;
  }
}

// Neighbour-mixed local rewards (reference common/reward.py:67-88): one thread per (agent, replica) instead of a loop
// over the agents inside k_stats; the adds of one agent's neighbours keep their order.
__global__ void k_local_rewards(DevNet N, DevState S, tsc_step_out_t out, int n_rb, int n_ablk) {
  int rb, ab;
  if (!map_block(n_rb, n_ablk, rb, ab)) return;
  const int r = rb * 64 + (int)threadIdx.x, a = ab * (int)blockDim.y + (int)threadIdx.y;
  if (r >= S.R || a >= N.n_agents || TSC_PARKED(S, r)) return;
  double rew = S.agent_local[IDX(a)];
  for (int k = N.agent_nbr_ptr[a]; k < N.agent_nbr_ptr[a + 1]; k++) rew += N.agent_nbr_w[k] * S.agent_local[IDX(N.agent_nbr[k])];
  if (out.local_reward_dev) out.local_reward_dev[(size_t)r * N.n_agents + a] = (float)rew;
  if (out.local_reward64_dev) out.local_reward64_dev[(size_t)r * N.n_agents + a] = rew;
}

// Network metrics, rewards, episode flags (reference backends/cityflow/metrics.py:24-260,
// common/reward.py:54-88,106-136, pytsc/__init__.py:58-62, simulator.py:20-30).
__global__ void k_stats(DevNet N, DevState S, DevCfg C, tsc_step_out_t out) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= S.R) return;
  if (TSC_PARKED(S, r)) { // (tsc_park: the replica's rows are not written; the caller is told)
    if (out.valid_dev) out.valid_dev[r] = 0;
    return;
  }
  const int A = N.n_agents;
  long long nq_all = S.nq_all[r], nveh_all = 0;
  S.nq_all[r] = 0;
  double dens = 0, nms = 0, tvs = 0;
  if (out.stats_dev) { // (k_net_sums, launched just before: the lane sums in lane order)
    dens = S.net_sums[IDX(0)];
    nms = S.net_sums[IDX(1)];
    tvs = S.net_sums[IDX(2)];
    nq_all = S.net_cnt[IDX(0)];
    nveh_all = S.net_cnt[IDX(1)];
  }
  double density = dens / N.n_lanes;
  double norm_mean_speed = nms / N.n_lanes;
  double mean_speed = nveh_all == 0 ? 0.0 : tvs / (double)nveh_all;
  double pressure = np_pairwise_sum_col(S.agent_pressure + r, (size_t)S.RS, A);
  int n_changed = 0;
#pragma unroll 16
  for (int a = 0; a < A; a++) n_changed += S.changed[IDX(a)] ? 1 : 0;
  double flick = (double)n_changed / (double)A;
  double g;
  if (C.reward_kind == 0) {
    g = 1e-6;
    g += C.flickering_coef * flick;
    g += (double)nq_all;
    g = -1 * g;
  } else {
    g = 1e-6;
    g -= C.flickering_coef * flick;
    g -= pressure;
  }
  if (out.global_reward_dev) out.global_reward_dev[r] = (float)g;
  if (out.global_reward64_dev) out.global_reward64_dev[r] = g;
  int step = S.t[r] + 1; // engine steps completed
  int sim_step = step - C.initial_wait_time;
  if (out.stats_dev) {
    double now = step * C.interval;
    // finished durations + (now - creation time) of every created, unfinished vehicle
    int set = S.set[r];
    int tt_i = step < N.sim_len + 1 ? step : N.sim_len + 1;
    long long created = N.created_cnt[(size_t)set * (N.sim_len + 2) + tt_i];
    long long created_ts = N.created_tsum[(size_t)set * (N.sim_len + 2) + tt_i];
    long long fin = S.fin_cnt[r];
    long long fin_dur = (long long)S.fin_dsum[r];   // sum of (finish step - creation step)
    long long fin_start = (long long)S.fin_ssum[r]; // sum of creation steps of finished vehicles
    // pooled = created, unfinished (waiting included); sum over them of (now - creation time).
    // With interval == 1 every term is an integer, so this equals the engine's f64 loop exactly.
    long long pooled = created - fin;
    long long pooled_wait = pooled * (long long)step - (created_ts - fin_start);
    long long n_all = fin + pooled;
    double avg = n_all == 0 ? 0.0 : ((double)fin_dur * C.interval + (double)pooled_wait * C.interval) / (double)n_all;
    double *s = out.stats_dev + (size_t)r * TSC_N_STATS;
    s[TSC_S_TIME_STEP] = now;
    s[TSC_S_AVG_TRAVEL_TIME] = avg;
    s[TSC_S_N_QUEUED] = (double)nq_all;
    s[TSC_S_MEAN_SPEED] = mean_speed;
    s[TSC_S_MEAN_DELAY] = 1 - norm_mean_speed;
    s[TSC_S_DENSITY] = density;
    s[TSC_S_PRESSURE] = pressure;
    s[TSC_S_NETWORK_FLOW] = density * norm_mean_speed;
  }
  if (out.done_dev) out.done_dev[r] = (unsigned char)(sim_step > 0 && sim_step % C.episode_limit == 0);
  if (out.terminated_dev) out.terminated_dev[r] = (unsigned char)(sim_step == C.sim_length);
  if (out.valid_dev) out.valid_dev[r] = 1; // (this family holds every drivable bumper to bumper: no replica ever stalls)
}

// --------------------------------------------------------------- k_snapshot
__global__ void k_snapshot(DevNet N, DevState S, int rep, int cap, int *count, int *drv, int *slot, double *dist,
                           double *speed, int *vid) {
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= N.n_drv) return;
  int r = rep;
  int hdr = S.cnt[IDX(d)];
  int n = (hdr & TSC_CNT_MASK) - ((hdr & TSC_NEW_BIT) ? 1 : 0);
  int sb = N.drv_slot_base[d];
  for (int i = 0; i < n; i++) {
    int k = atomicAdd(count, 1);
    if (k < cap) {
      drv[k] = d;
      slot[k] = i;
      const double2 vi = S.ds[DIDX(sb + i)];
      dist[k] = vi.x;
      speed[k] = vi.y;
      vid[k] = TSC_TAG_VID(S.tag[SIDX(sb + i)]);
    }
  }
}

__global__ void k_park_list(DevState S, const unsigned char *mask) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < S.R && mask[r]) S.parked[r] = 1;
}

// Uniformly random allowed action per (replica, agent): one launch instead of a chain of
// tensor ops (reference RandomController, controllers.py:245-268).
__global__ void k_random_valid(DevState S, DevCfg C, int A, const unsigned char *mask, unsigned long long seed,
                               unsigned long long draw, int *actions) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S.R * A) return;
  const unsigned char *m = mask + (size_t)i * C.n_actions;
  int allowed = 0;
  for (int k = 0; k < C.n_actions; k++) allowed += m[k] != 0;
  // splitmix64 of (seed, draw index, GLOBAL replica * A + agent): the same draw whatever the sharding (replica_base)
  unsigned long long z = seed + 0x9e3779b97f4a7c15ull * (draw + 1) +
                         0xbf58476d1ce4e5b9ull * ((unsigned long long)C.replica_base * (unsigned long long)A + (unsigned long long)(i + 1));
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  int pick = allowed > 0 ? (int)(((z >> 32) * (unsigned long long)allowed) >> 32) : 0;
  int act = 0;
  for (int k = 0; k < C.n_actions; k++)
    if (m[k] != 0 && pick-- == 0) {
      act = k;
      break;
    }
  actions[i] = act;
}

// The reference's rule-based controllers (pytsc/controllers/controllers.py:26-268) for every (replica, signal) in one
// launch, from what the step just wrote: phase [R, A, 3] (index, time on it, changed) and the per-phase scores [R, A,
// TSC_MAX_PHASES, 3] (stopped vehicles, vehicles, pressure: tsc_step_out_t::phase_scores_dev).  Allowable phases:
// TLSFreePhaseSelectLogic / TLSRoundRobinPhaseSelectLogic (common/traffic_signal.py:329-404), as the action masks of the
// step.  kind / p0..p2: include/pytsc_amd.h, TSC_CTRL_*.
__global__ void k_controller(DevNet N, DevState S, DevCfg C, int kind, int p0, int p1, int p2, unsigned long long seed,
                             unsigned long long draw, const int *phase, const short *scores, int *actions) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int A = N.n_agents;
  if (i >= S.R * A) return;
  const int a = i % A;
  const int cur = phase[(size_t)i * 3], top = phase[(size_t)i * 3 + 1];
  const int np_ = N.agent_n_phases[a];
  const int *green = N.agent_phase_green + a * TSC_MAX_AGENT_PHASES;
  const int nxt = cur + 1 == np_ ? 0 : cur + 1;
  const bool is_green = green[cur] != 0;
  // allowable phase switches (bit p): green: t < min {cur}; min <= t < max {cur, next}; t == max {next}; yellow: free --
  // every green phase but the one just left; round robin -- the next one
  unsigned int allow = 0;
  if (is_green) {
    const int mn = N.agent_phase_min[a * TSC_MAX_AGENT_PHASES + cur], mx = N.agent_phase_max[a * TSC_MAX_AGENT_PHASES + cur];
    if (top < mn) allow = 1u << cur;
    else if (top < mx) allow = (1u << cur) | (1u << nxt);
    else if (top == mx) allow = 1u << nxt;
  } else if (C.round_robin) {
    allow = 1u << nxt;
  } else {
    for (int k = 0; k < np_; k++)
      if (green[k] && k != cur - 1) allow |= 1u << k;
  }
  // splitmix64 of (seed, call, GLOBAL replica * A + agent), as k_random_valid
  unsigned long long z = seed + 0x9e3779b97f4a7c15ull * (draw + 1) +
                         0xbf58476d1ce4e5b9ull * ((unsigned long long)C.replica_base * (unsigned long long)A + (unsigned long long)(i + 1));
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  const short *sc = scores ? scores + (size_t)i * (TSC_MAX_PHASES * 3) : nullptr;
  int target = nxt;
  if (kind == TSC_CTRL_FIXED_TIME) {
    target = (is_green && top < p0) ? cur : nxt; // (controllers.py:39-54)
  } else if (kind == TSC_CTRL_GREEDY || kind == TSC_CTRL_MAX_PRESSURE || kind == TSC_CTRL_RANDOM) {
    if (is_green || kind == TSC_CTRL_RANDOM) {
      // the allowable phases with the largest score (random: every allowable phase), one of them uniformly at random
      const int col = kind == TSC_CTRL_GREEDY ? 0 : 2;
      int best = -1;
      unsigned int ties = 0;
      for (int k = 0; k < np_ && k < TSC_MAX_PHASES; k++)
        if ((allow >> k) & 1u) {
          const int v = kind == TSC_CTRL_RANDOM ? 0 : (int)sc[k * 3 + col];
          if (v > best) {
            best = v;
            ties = 1u << k;
          } else if (v == best) ties |= 1u << k;
        }
      const int n_t = __builtin_popcount(ties);
      if (n_t > 0) {
        int pick = (int)(((z >> 32) * (unsigned long long)n_t) >> 32);
        unsigned int t = ties;
        while (pick-- > 0) t &= t - 1u;
        target = __builtin_ctz(t);
      }
    }
  } else if (kind == TSC_CTRL_SOTL) { // (controllers.py:200-242; p0 = theta, p1 = mu, p2 = phi_min)
    if ((allow >> cur) & 1u) {
      int ng = nxt + 1 == np_ ? 0 : nxt + 1; // next green phase index: (cur + 2) % n_phases
      const int red = ng < TSC_MAX_PHASES ? (int)sc[ng * 3 + 1] : 0, grn = cur < TSC_MAX_PHASES ? (int)sc[cur * 3 + 1] : 0;
      target = (top >= p2 && !(grn > 0 && grn < p1) && red >= p0) ? nxt : cur;
    } else target = nxt;
  }
  actions[i] = C.action_space == TSC_ACT_PHASE_SWITCH ? (target != cur ? 1 : 0) : target;
}

__global__ void k_count_vehicles(DevNet N, DevState S, unsigned long long *total) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= S.R) return;
  unsigned long long c = 0;
  for (int d = 0; d < N.n_drv; d++) {
    int hdr = S.cnt[IDX(d)];
    c += (hdr & TSC_CNT_MASK) - ((hdr & TSC_NEW_BIT) ? 1 : 0);
  }
  atomicAdd(total, c);
}

#include "tsc_resident.inc"

// Action log of a handle that may migrate from the resident to the list family (tsc_api.inc): one byte per action, written
// by kr_step's launch over all replicas (tsc_resident.inc); read back here.
__global__ void k_unlog_actions(int *dst, const signed char *src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (int)src[i];
}

// Vehicle bookkeeping per replica (conservation check K5): entered the network so far, running,
// finished.  A vehicle inserted for the NEXT sub-step counts in none of the three.
__global__ void k_replica_counts(DevNet N, DevState S, int *out) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= S.R) return;
  int entered = 0, running = 0;
  for (int d = 0; d < N.n_drv; d++) {
    int hdr = S.cnt[IDX(d)];
    int nw = (hdr & TSC_NEW_BIT) ? 1 : 0;
    running += (hdr & TSC_CNT_MASK) - nw;
    if (d < N.n_lanes) entered += S.spawn_cur[IDX(d)] - nw;
  }
  out[r * 3 + 0] = entered;
  out[r * 3 + 1] = running;
  out[r * 3 + 2] = S.fin_cnt[r];
}

#include "tsc_api.inc"
