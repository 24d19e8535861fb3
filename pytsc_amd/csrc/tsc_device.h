// Device-side views of the static tables and the dynamic state (gfx950 only).
//
// HBM layout: structure-of-arrays with the REPLICA index
// fastest.  A wavefront is 64 consecutive replicas of ONE drivable (or one
// agent), so every state access `arr[row * RS + r]` is a 256/512-byte coalesced
// segment and every static-table access is wave-uniform (scalar loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/pytsc_amd.h"

// The header word of a drivable (per replica): vehicles on it (bits 0-15), this sub-step's leavers appended to its outbox
// (bits 16-23, k_exec's atomic adds; k_commit keeps them through its own write of the word -- the target drivables of the
// same launch read them -- and the next k_enum clears them), TSC_NEW_BIT: its tail was inserted in this step.
// (Round 4: the outbox count was a word of its own -- a store per drivable and replica in k_enum, a row more to read in k_commit.)
#define TSC_CNT_MASK 0xffff
#define TSC_OB_ONE 0x10000
#define TSC_OB_BITS 0xff0000
#define TSC_OB_OF(hdr) (((hdr) >> 16) & 0xff)
// bit 24: a vehicle is on its way here in this sub-step (k_exec sets it in the TARGET's word when it appends a leaver to its
// own drivable's outbox; k_commit's quiet test then needs no look at the feeders; its write of the word clears it)
#define TSC_ARR_BIT 0x1000000
#define TSC_NEW_BIT 0x40000000
#define TSC_TAG_VID(tag) ((tag) & 0xffffff)
#define TSC_TAG_SEQ(tag) (((uint32_t)(tag)) >> 24)
#define TSC_MAKE_TAG(vid, seq) ((int32_t)(((uint32_t)(vid)) | (((uint32_t)(seq)) << 24)))
#define TSC_F_MOVED 1
#define TSC_F_ENDED 2
#define TSC_OB_MAX 3

// One cross of a lane-link, 32 bytes so the wave-uniform scan fetches it with one scalar load.
// Device-side vehicle template rows: the TSC_TMPL_STRIDE fields of the C-ABI table followed by
// constants derived from them at create with the very f64 operations the kernels used to
// spend per vehicle (IEEE division is correctly rounded on both sides: same bits).
#define TSC_DT_STRIDE 20
#define TSC_DT_HA_MAX 12   /* 0.5 / maxNegAcc */
#define TSC_DT_IA_MAX 13   /* 0.5 / (0.5 / maxNegAcc) */
#define TSC_DT_HA_USUAL 14 /* 0.5 / usualNegAcc */
#define TSC_DT_IA_USUAL 15 /* 0.5 / (0.5 / usualNegAcc) */
#define TSC_DT_LIMIT 16    /* maxSpeed * maxSpeed / usualNegAcc / 2 + maxSpeed * interval * 2 (look-ahead limit) */
#define TSC_DT_HWDEN 17    /* headway + interval / 2 */

#ifndef TSC_ROWPACK
#define TSC_ROWPACK 2 /* rows of {dist, speed} stored side by side per replica (DIDX) */
#endif

// Wave-uniform constants of one work list (k_enum -> k_exec), 128 bytes: everything a
// batch needs about its drivable and the lane-link it is heading for, in one scalar load.
struct ListRec {
  int d, sb, pick, link;        // drivable, its slot base, out-link index (TSC_MAXQ: plain followers), lane-link or -1
  int lk, el, ag, link_type;    // link - n_lanes, end lane of the link, agent, turn type
  int olk[4];                   // lanes: the lane's out-links (-2: none)
  int cross_b, cross_e;         // cross records of `link`
  unsigned int avail_mask;      // raw phases in which `link` is green
  int is_link;
  double L, drvMax, my_len, first_cross_x; // drivable length / speed limit, length of `link`, its first cross
  long long item_off;           // start of the list's items in wl_items
  unsigned long long conflict;  // links of the intersection crossing `link`
  unsigned int avail4[4];       // lanes: green masks of the four out-links (olk order)
};

// What k_exec needs of an intersection, per replica, in one 32-byte record: the raw phase
// (k_phase) and the three cross-candidate masks (k_notify).
struct AgentRec {
  unsigned long long oe, ov, os;
  int raw, pad;
};

// A vehicle leaving its drivable in this sub-step, as k_exec hands it to k_commit.
struct ObxRec {
  double dist, speed;
  int tag, aux, blk, slot;
};

// What k_commit needs to keep the cross-candidate masks of the two intersections a lane touches up to date, in one
// 64-byte scalar load: the lane's feeder lane-links (it is their end lane) and its out-links (it is their start lane) as
// local-index masks of their intersections, and link by link (drivable index, -1: none; local index in byte q of *_loc).
struct LaneMaskRec {
  unsigned long long fm, om; // feeder links / out-links as bits of their intersection's local indices
  int ag_in, ag_out;         // the intersections (-1: the lane has no such links)
  int fid[4], oid[4];
  unsigned int floc, oloc;
};

struct CrossRec {
  double x;   // distance of the cross on this link
  double xo;  // distance of the cross on the other link
  int other;  // the other lane-link (drivable index)
  int meta;   // local index of the other link (bits 0-7) | its road-link type << 8
  int pe;     // entry of the same cross in the other link's list
  int pad;
};

// Pointers inside structs that a kernel reads through a pointer (RParams of the resident family) carry the
// global address space in device code: a pointer loaded from memory would otherwise be generic and every access
// through it a flat_* instruction.  Host code sees plain pointers of the same size.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define TSC_G __attribute__((address_space(1)))
#else
#define TSC_G
#endif
template <typename T>
using GP = const T TSC_G *; // read-only table in global memory

struct DevNet {
  int n_lanes, n_links, n_drv, n_agents, n_routes, n_tmpl, single_tmpl;
  int max_links_per_agent;
  GP<double> drv_length, drv_max_speed;
  GP<int> drv_cap, drv_slot_base;
  GP<int> lane_out_ptr, lane_out_links, lane_in_ptr, lane_in_links;
  GP<double> lane_len_pytsc;
  GP<double> lane_slots; // lane_len_pytsc / veh_size_min_gap: the divisor of a lane's occupancy
  // correctly rounded reciprocals of a lane's two divisors -- lane_slots and the speed limit -- for div_c (k_net_sums);
  // recip_ok: every lane's divisors meet the theorem's conditions (else the kernel divides)
  GP<double> lane_rslots, lane_rmax;
  int recip_ok;
  GP<int> lane_bins, lane_static;
  GP<LaneMaskRec> lane_mask; // [n_lanes]; a lane with more than four feeder links has fid[0] = -2 (k_commit then walks the tables)
  GP<uint32_t> lane_static_pk; // the nine static features of a lane as 2-bit symbols (two's complement: -1, 0, 1), feature j in bits 2j, 2j + 1
  GP<int> lane_is_entry; // 1: some flow set spawns vehicles on this lane
  GP<int> link_start_lane, link_end_lane, link_type, link_agent, link_local;
  GP<int2> link_lane_local; // per lane-link: position of its start lane among its intersection's incoming lanes, of its end lane among the outgoing ones
  GP<uint32_t> link_avail_mask;
  GP<unsigned long long> link_conflict; // local-index mask of links crossing this link
  GP<unsigned long long> agent_phase_avail; // [n_agents*32]: local-index mask of links available in raw phase p
  GP<int> link_cross_ptr, cross_other, cross_peer; // cross_peer: entry of the same cross in the other link's list
  GP<double> cross_dist_self, cross_dist_other;
  GP<CrossRec> cross_rec; // [n_cross + 4], padded for the software-pipelined scan
  GP<int> agent_link_ptr, agent_links; // CSR: links of an agent in local order
  GP<int> agent_n_phases, agent_phase_raw, agent_phase_green, agent_phase_min, agent_phase_max;
  GP<int> agent_inc_ptr, agent_inc_lanes, agent_out_ptr, agent_out_lanes;
  GP<int> agent_nbr_ptr, agent_nbr;
  GP<double> agent_nbr_w;
  GP<int> route_ptr, route_drv;
  GP<int2> veh_rrange; // per vehicle (all flow sets): [begin, end) of its route in route_drv
  GP<double> tmpl;
  // flows
  int n_sets, sim_len;
  GP<int> set_veh_ptr, veh_route, veh_start, veh_tmpl, spawn_ptr, spawn_vid, created_cnt;
  GP<long long> created_tsum;
  GP<int> veh_prio; // tsc_flows_t::veh_priority (null: the lower vehicle id goes first)
  GP<uint4> agent_moves; // tsc_net_t::agent_phase_moves ([n_agents * TSC_MAX_AGENT_PHASES]; null: not provided)
};

// counter-based normal draws of the device-side observation noise (tsc_config_t::obs_noise_std): murmur3 finaliser +
// Box-Muller, keyed on (seed, global replica, engine step) -> key0, then (signal, lane, ordinal of the surviving bin) -- the
// same key gives the same draw on either kernel family
__device__ __forceinline__ unsigned long long r_fmix64(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ float r_noise_normal(unsigned long long key0, unsigned int a, unsigned int lane, unsigned int ordinal) {
  const unsigned long long h = r_fmix64(key0 ^ (((unsigned long long)a << 40) | ((unsigned long long)lane << 20) | ordinal) * 0x9E3779B97F4A7C15ull);
  const float u1 = ((float)(unsigned int)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); // (0, 1): 24 bits each
  const float u2 = ((float)(unsigned int)(h & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
__device__ __forceinline__ unsigned long long tsc_noise_key0(unsigned long long seed, int global_replica, int t_done) {
  return r_fmix64(seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)(unsigned int)global_replica << 32) | (unsigned int)t_done));
}
#define TSC_PARKED(S, r) ((S).any_parked && (S).parked[(r)])
// last tie-break of Cross::canPass between vehicles `vid` and `fvid` of the flow set at `vbase`: true = `vid` goes first
__device__ __forceinline__ bool tsc_goes_first(const DevNet &N, int vbase, int vid, int fvid) {
  if (N.veh_prio) {
    const int p1 = N.veh_prio[vbase + vid], p2 = N.veh_prio[vbase + fvid];
    if (p1 != p2) return p1 > p2;
  }
  return vid < fvid;
}

struct DevState {
  int R, RS; // replicas, row stride (R rounded up to 64)
  // tsc_park on the list family (round 5): parked[r] != 0 -- every kernel that walks replica r's state treats its lane as it
  // treats a lane beyond R: nothing moves, nothing is written, valid_dev = 0 -- until a reset that selects r.  any_parked: the
  // handle has parked replicas at all (a kernel argument: handles that never park pay no load for the test).
  int any_parked;
  unsigned char *parked; // [RS]
  int n_slots; // total vehicle slots per replica (rows of the tiled slot arrays)
  // per slot [S_total * RS]
  // One vehicle = {dist, speed} (16 bytes, ping-pong) + {aux, blk} (8 bytes) + tag (4 bytes), the
  // last two in place: the walk of k_enum reads the first two, a settled vehicle costs one store.
  double2 *ds;  // committed half (read by k_notify / k_enum / k_exec / k_measure)
  double2 *nds; // the other half: written by k_enum / k_exec, patched by k_commit, then swapped in
  int2 *ab;     // x: aux (next drivable / link entry time), y: blocker word (TSC_BLK_*)
  int *tag;
  // per drivable [n_drv * RS]
  int *cnt, *tail_tag, *tail_prev;
  // outbox of this sub-step's leavers: how many is in the header word cnt (TSC_OB_OF), records [n_drv * TSC_OB_MAX * RS]
  ObxRec *obx; // [drivable][TSC_OB_MAX][replica]: this sub-step's leavers
  double2 *tail;  // {dist, speed} of the tail of the committed state (the tail before a spawn)
  double2 *ntail; // next half: written by k_enum / k_exec (quiet drivables) / k_commit, then swapped in
  // per lane [n_lanes * RS]
  int *spawn_cur, *lane_n, *lane_q;
  int *fz; // [lane][replica] k_enum's standing block: fa | fb << 16 (slots (fa, fb] are skipped)
  double *lane_sum;
  unsigned long long *lane_in_bits; // 3 predicate bits per incoming-window bin (visibility <= 16)
  float *lane_in_val; // [n_lanes * visibility * RS] (observation noise only, else null): the VALUE of every bin of the
                      // incoming window, x = -1 + sum(1 + s / s_max) (retriever.py:20-52), what the noise is added to
  unsigned int *lane_out_bits;
  // per agent [n_agents * RS]
  int *cur_idx, *top, *changed;
  AgentRec *arec; // raw phase + cross-candidate masks, read together by k_exec
  double *agent_local, *agent_pressure;
  double *net_sums; // [3 * RS]: density, normalised mean speed, speed x n summed over all lanes in lane order (k_net_sums)
  int *net_cnt;     // [2 * RS]: queued vehicles, vehicles on all lanes
  // per replica [RS]
  int *t, *set, *restarts, *fin_cnt;
  int *nq_all; // queued vehicles on ALL lanes (Q6), accumulated by k_measure, consumed + zeroed by k_stats
  unsigned long long *fin_dsum, *fin_ssum; // sums over finished vehicles: duration, creation step
  // per vid [max_veh * RS]: committed blocker of a running vehicle, ping-pong halves
  int *bbv, *nbbv;
  // work lists of k_enum for k_exec (all replicas): wl_nb[0] = number of batches (wl_nb[1] unused)
  // followed in the same allocation by wl_cnt[n_lists]; both zeroed before every k_enum.
  int *wl_nb;
  int n_lists;
  int *wl_cnt;
  const ListRec *wl_rec;   // [n_lists]
  unsigned int *wl_items;
  int2 *wl_desc;           // batch table of k_plan: (list, batch)
  int *errors; // [4]: slot overflow, outbox overflow, bad action, other
};

struct DevCfg {
  int delta_time, yellow_time, action_space, round_robin, visibility, sim_length, episode_limit;
  int initial_wait_time, set_offset, sequential_sets, reward_kind, n_actions, obs_size, state_size;
  int ablate; // diagnostic builds only (TSC_ABLATE env): skip parts of k_enum / k_exec to price them; 0 in every real run
  // resident family, runtime knobs of its data-dependent paths (tsc_config_t::resident_force / _task_capacity / _margin):
  int r_force;    // bit 0: every sub-step takes the two-phase hand-off path; bit 1: the measure phase walks the lists
  int r_task_cap; // cross tasks per sub-step held in LDS (<= R_TASKS); beyond: the vehicle scans its crosses serially
  int r_margin;   // records of headroom below which a replica asks for the migration to the list family
  int r_small_up;   // two-class handles: the small class's capacity less R_CLS_MARGIN: a replica that needs more is the full-size class's
  int r_small_down; // two-class handles: a replica of the full-size class that needs fewer records than this goes back to the
                    // small class (its capacity less twice R_CLS_MARGIN); 0: the handle has no small class
  double interval, veh_size_min_gap, flickering_coef;
  // device-side observation noise (tsc_config_t::obs_noise_std / _seed; resident family): 0 = off
  float obs_noise_std;
  unsigned long long obs_noise_seed;
  int replica_base; // tsc_config_t::replica_base: global index of replica 0 (keys of the counter-based draws)
};
