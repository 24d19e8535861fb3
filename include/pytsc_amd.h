/*
 * pytsc_amd.h — C-ABI of the MI355X-native batched traffic-signal environment.
 *
 * This is the drop-in boundary for the hot path `TrafficSignalNetwork.step()`
 * (reference: pytsc/__init__.py:178-182).  The reference has no FFI of its own:
 * its boundary is a Python class registry (pytsc/__init__.py:11-14,
 * pytsc/backends/cityflow/__init__.py:18-26) on top of the pybind module
 * `cityflow.Engine`.  The entry points below are what a `pytsc.backends.<name>`
 * backend binds (via ctypes) instead of `cityflow.Engine`:
 *
 *   tsc_create        <- cityflow.Engine(config_file, thread_num)
 *                        (pytsc/backends/cityflow/simulator.py:64-78) for R replicas
 *   tsc_reset         <- Engine.reset() + new Engine (simulator.py:91-95,
 *                        pytsc/__init__.py:164-176)
 *   tsc_step          <- action_space.apply + 5 x Engine.next_step +
 *                        retrieve_step_measurements + update_stats + get_reward
 *                        (pytsc/__init__.py:178-182; actions.py:99-108,144-158;
 *                        simulator.py:80-89; retriever.py:54-112;
 *                        backends/cityflow/traffic_signal.py:101-141)
 *                        and the getters the callers pull right after:
 *                        get_observations / get_state / get_rewards /
 *                        get_action_mask (observations.py:140-160,183-205;
 *                        reward.py:54-88; actions.py:119-131,169-188)
 *   tsc_lane_detail   <- Engine.get_lane_vehicles / get_vehicle_info based
 *                        per-lane position matrices (retriever.py:20-52) for the
 *                        single-replica facade objects the rule-based
 *                        controllers read (controllers/controllers.py:93-242)
 *   tsc_snapshot      <- Engine.get_lane_vehicles/get_vehicle_speed/
 *                        get_vehicle_info dumps (debug / parity)
 *   tsc_destroy       <- Engine destructor
 *
 * Conventions: plain pointers and sizes only; `*_dev` pointers are caller-owned
 * DEVICE buffers (e.g. torch tensors' data_ptr()); tables are HOST pointers
 * copied at create time; `stream` is a hipStream_t passed as void* (NULL = the
 * null stream); every call returns 0 on success or a negative TSC_E* code and
 * never blocks the host except tsc_create/tsc_destroy/tsc_snapshot/tsc_errors.
 * No internal host threads; one handle per GPU; a handle is driven from ONE
 * stream at a time (its launch constants -- among them the output pointers of
 * the step in flight -- are updated in the order of that stream: a caller that
 * changes stream synchronises the old one first).
 *
 * The same table structs are consumed by the CPU oracle (oracle/tsc_oracle.c includes
 * this header; oracle/pyoracle.py binds it), which is test infrastructure only.
 */
#ifndef PYTSC_AMD_H
#define PYTSC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSC_ABI_VERSION 5

/* error codes */
#define TSC_OK 0
#define TSC_EINVAL (-1)   /* bad argument / inconsistent tables            */
#define TSC_ENODEV (-2)   /* no usable GPU / HIP failure                   */
#define TSC_ENOMEM (-3)
#define TSC_EUNSUP (-4)   /* table uses a feature the HIP path rejects     */

/* vehicle template: doubles per template (flow JSON "vehicle" object) */
#define TSC_TMPL_STRIDE 12
#define TSC_T_LEN 0
#define TSC_T_WIDTH 1
#define TSC_T_MAXPOSACC 2
#define TSC_T_MAXNEGACC 3
#define TSC_T_USUALPOSACC 4
#define TSC_T_USUALNEGACC 5
#define TSC_T_MINGAP 6
#define TSC_T_MAXSPEED 7
#define TSC_T_HEADWAY 8
#define TSC_T_YIELDDIST 9
#define TSC_T_TURNSPEED 10
#define TSC_T_APPROACH 11 /* maxSpeed^2/(2 usualNegAcc) + 2 maxSpeed dt */

/* road-link types, numeric order is the yielding priority */
#define TSC_LINK_RIGHT 1
#define TSC_LINK_LEFT 2
#define TSC_LINK_STRAIGHT 3

/* observation geometry (reference: common/observations.py:59-63,175-181,207-213) */
#define TSC_MAX_CTRL_LANES 16
#define TSC_MAX_PHASES 20
#define TSC_LANE_STATIC 9
#define TSC_MAX_AGENT_PHASES 32 /* PyTSC phase-list entries per agent (table limit) */

/* action spaces (reference: common/actions.py:87,134) */
#define TSC_ACT_PHASE_SELECTION 0
#define TSC_ACT_PHASE_SWITCH 1

/* Kernel families (same results, bit for bit):
 *   LIST      state in HBM, replica-fastest arrays, one wavefront = 64 replicas of one drivable,
 *             dense work lists across replicas, 26 kernel launches + 5 counter memsets per env step (k_phase;
 *             5 x {k_enum, k_plan, k_exec, k_commit}; k_measure, k_agent, k_net_sums, k_local_rewards,
 *             k_stats); any network size.
 *   RESIDENT  one workgroup per replica, the replica's whole state in LDS across the delta_time
 *             sub-steps and the measurement; networks whose replica fits on chip (grids up to
 *             about 5x5, Jinan 3x4).  tsc_create returns TSC_EUNSUP if the tables do not fit it.
 *             ABI 5: two CLASSES of workgroups step the replicas out of the same per-replica blobs
 *             -- those that need few vehicle records in 512-thread workgroups with an LDS image of
 *             at most 80 KB, two per CU (1.5-1.7x the rate per replica), the others in 768-thread
 *             workgroups with the whole CU's LDS; one launch per class and env step; a replica
 *             changes class by a byte the kernel itself sets (tsc_class_counts).
 *   AUTO      RESIDENT where it fits, else LIST.  A replica of an AUTO handle on RESIDENT that comes within resident_margin
 *             records of its capacity is PROMOTED to the handle's spill pool: its state is copied into a blob with room for
 *             every drivable bumper to bumper (up to 16320 records) and the same kernel steps it from there with its
 *             working image in global memory instead of LDS -- several times slower for that replica, bit-identical
 *             results, the other replicas untouched.  Once more than resident_spill_limit replicas have asked, the whole
 *             handle moves to LIST instead, as follows.  An AUTO handle on RESIDENT keeps the LIST arrays as a stand-by and
 *             logs resets / actions on the device: when a replica comes within resident_margin vehicle records of
 *             its capacity the handle moves to LIST (same results): while no replica has stalled, by writing the state of
 *             every replica across into the LIST arrays (round 5: a millisecond); with a stalled replica -- its state is
 *             steps old -- by replaying the log.  The flag travels to the host
 *             behind every step and is read, without synchronising, by the next tsc_step; a caller that enqueues steps
 *             without waiting is held back so that it never runs more than 16 steps ahead of the flag.  Should a replica
 *             run out of records all the same (more than the margin gained within those steps), it STALLS: its state
 *             stays that of its last completed step, its rows in the output buffers are not written (valid_dev[r] = 0
 *             tells the caller which replicas) until the host gets to it: promoted, it is caught up from the actions of
 *             the last 32 steps, which the handle keeps; or the migration's replay repairs everything -- either way the
 *             state afterwards is exact and tsc_errors stays 0 (the log holds 1024 env steps and 64 partial restarts since
 *             the last full restart; beyond that there is no way back and a stall raises the error counter, as it does
 *             on a handle forced to RESIDENT).  A full restart (tsc_reset / tsc_reset_sets with no mask or with a mask that selects every
 *             replica) starts the log afresh and takes a migrated handle back to RESIDENT; to tell, such a handle
 *             copies the mask to the host and synchronises the stream (restarts are rare).  A handle forced to
 *             RESIDENT has no stand-by: a spawn that finds no free record raises the error counter (tsc_errors). */
#define TSC_FAMILY_AUTO 0
#define TSC_FAMILY_LIST 1
#define TSC_FAMILY_RESIDENT 2

/* stats columns of tsc_step's `stats` output (reference: backends/cityflow/metrics.py:221-260) */
#define TSC_N_STATS 8
#define TSC_S_TIME_STEP 0
#define TSC_S_AVG_TRAVEL_TIME 1
#define TSC_S_N_QUEUED 2
#define TSC_S_MEAN_SPEED 3
#define TSC_S_MEAN_DELAY 4
#define TSC_S_DENSITY 5
#define TSC_S_PRESSURE 6
#define TSC_S_NETWORK_FLOW 7

/* per-agent stats columns (reference: backends/cityflow/traffic_signal.py:101-141) */
#define TSC_N_AGENT_STATS 8
#define TSC_A_N_QUEUED 0
#define TSC_A_OCCUPANCY 1
#define TSC_A_MEAN_SPEED 2
#define TSC_A_MEAN_DELAY 3
#define TSC_A_OUT_OCCUPANCY 4
#define TSC_A_PRESSURE 5
#define TSC_A_NORM_TIME_ON_PHASE 6
#define TSC_A_SIM_STEP_NORM 7

/*
 * Static network tables (shared by all replicas).  Drivable index space:
 * lanes first, [0, n_lanes), in *sorted lane-id order* (the order of
 * std::map keys that cityflow's getters return and PyTSC iterates,
 * retriever.py:66, network_parser.py:104-118); lane-links follow,
 * [n_lanes, n_lanes + n_links), in roadnet-file order (intersection, roadLink,
 * laneLink).  Agents are in roadnet-file intersection order
 * (network_parser.py:48-51).
 */
typedef struct tsc_net {
  int32_t n_lanes;
  int32_t n_links;
  int32_t n_agents;
  int32_t n_routes;
  int32_t n_tmpl;
  int32_t n_raw_phases_max;

  /* drivables [n_lanes + n_links] */
  const double *drv_length;     /* engine length: trimmed lane / polyline of link */
  const double *drv_max_speed;  /* lane maxSpeed; 10000 for lane-links            */
  const int32_t *drv_cap;       /* slot capacity                                  */

  /* lanes [n_lanes] */
  const int32_t *lane_out_ptr;  /* [n_lanes+1] CSR: lane-links leaving the lane   */
  const int32_t *lane_out_links;/*   drivable indices, roadnet-file order         */
  const int32_t *lane_in_ptr;   /* [n_lanes+1] CSR: lane-links ending in the lane */
  const int32_t *lane_in_links;
  const double *lane_len_pytsc; /* centre-to-centre length (network_parser.py:325-352) */
  const int32_t *lane_bins;     /* int(len_pytsc / veh_size_min_gap)              */
  const int32_t *lane_static;   /* [n_lanes*9] truncated static features (Q1)     */

  /* lane-links [n_links] (index = drivable index - n_lanes) */
  const int32_t *link_start_lane;
  const int32_t *link_end_lane;
  const int32_t *link_type;       /* TSC_LINK_*                                  */
  const int32_t *link_agent;      /* owning agent index                          */
  const uint32_t *link_avail_mask;/* bit p set: available in raw light-phase p   */
  const int32_t *link_cross_ptr;  /* [n_links+1] CSR, sorted by cross_dist_self  */
  const int32_t *cross_other;     /* other lane-link (drivable index)            */
  const double *cross_dist_self;  /* distance of the cross on this link          */
  const double *cross_dist_other; /* distance of the cross on the other link     */

  /* agents [n_agents] */
  const int32_t *agent_n_phases;  /* PyTSC phase-list length                      */
  const int32_t *agent_phase_raw; /* [n_agents*TSC_MAX_AGENT_PHASES] raw phase id */
  const int32_t *agent_phase_green;/* [..] 1 = green index, 0 = yellow            */
  const int32_t *agent_phase_min; /* [..] min_time of that entry's raw phase      */
  const int32_t *agent_phase_max; /* [..] max_time                                */
  const int32_t *agent_inc_ptr;   /* [n_agents+1] CSR incoming lanes (sorted ids) */
  const int32_t *agent_inc_lanes;
  const int32_t *agent_out_ptr;   /* [n_agents+1] CSR outgoing lanes (sorted ids) */
  const int32_t *agent_out_lanes;
  const int32_t *agent_nbr_ptr;   /* [n_agents+1] CSR k-hop neighbours, k-major   */
  const int32_t *agent_nbr;       /*   neighbour agent index                      */
  const double *agent_nbr_w;      /*   gamma**k                                   */

  /* routes */
  const int32_t *route_ptr;       /* [n_routes+1] CSR */
  const int32_t *route_drv;       /* lane, link, lane, ..., lane */

  /* vehicle templates [n_tmpl*TSC_TMPL_STRIDE] */
  const double *tmpl;

  /* ABI 5 -- the movements of every phase, for the per-phase scores of the rule-based controllers
   * (tsc_step_out_t::phase_scores_dev; reference controllers/controllers.py:93-112,153-178,227-242 walk
   * config["phase_to_inc_out_lanes"][phase]): [n_agents * TSC_MAX_AGENT_PHASES * 4] words, per (agent, PyTSC phase index):
   *   word 0      bit k set: the agent's k-th incoming lane (agent_inc_lanes order) is a key of the phase's dict
   *   words 1, 2  sixteen 4-bit fields: field k = index, among the agent's outgoing lanes (agent_out_lanes order), of the
   *               LAST outgoing lane listed for incoming lane k (MaxPressureController counts only that one: :166-176)
   *   word 3      0
   * NULL: the host did not provide it (an agent with more than 16 incoming or outgoing lanes): phase_scores_dev and
   * tsc_controller_actions then return TSC_EUNSUP. */
  const uint32_t *agent_phase_moves;
} tsc_net_t;

/*
 * Flow sets: F alternative demand files; replica r uses set
 * (set_offset + r) % n_sets after each (re)start, advancing per restart when
 * `sequential` (reference: backends/cityflow/config.py:63-76).  A vehicle id
 * ("vid") is the rank of the vehicle in creation order (creation step, flow
 * index, count) inside its set.
 */
typedef struct tsc_flows {
  int32_t n_sets;
  int32_t sim_len;               /* engine steps covered by the prefix tables     */
  const int32_t *set_veh_ptr;    /* [n_sets+1] offsets into veh_*                 */
  const int32_t *veh_route;
  const int32_t *veh_start;      /* creation step                                 */
  const int32_t *veh_tmpl;
  const int32_t *spawn_ptr;      /* [n_sets*n_lanes+1] CSR over (set, lane)       */
  const int32_t *spawn_vid;      /*   vids waiting order (creation order)         */
  const int32_t *created_cnt;    /* [n_sets*(sim_len+2)] vehicles created at steps < t */
  const int64_t *created_tsum;   /* [n_sets*(sim_len+2)] sum of their creation times  */
  /* Last tie-break of Cross::canPass (equal road-link types, equal steps to the cross, equal link-entry times, equal
   * distances): CityFlow compares the vehicles' `priority`, drawn from the engine's std::mt19937(seed) when a vehicle is
   * created.  NULL (the default): the lower vehicle id goes first -- the documented deviation of rounds 1-3.  Else
   * [n vehicles, all sets] priorities, the HIGHER goes first (equal: the lower id): the host's restatement of that stream
   * (pytsc_amd/flows.py::mt19937_priorities, config key `priority_rule: mt19937`) -- a hypothesis about an engine that is
   * not here, kept behind this switch so that a recorded CityFlow trace can tell the two rules apart
   * (tools/capture_cityflow_golden.py, tests/test_cityflow_pin.py). */
  const int32_t *veh_priority;
} tsc_flows_t;

typedef struct tsc_config {
  int32_t n_replicas;
  int32_t delta_time;            /* engine sub-steps per env step (5)             */
  int32_t yellow_time;           /* == delta_time (backends/cityflow/config.py:58-61) */
  int32_t action_space;          /* TSC_ACT_*                                     */
  int32_t round_robin;           /* TLSRoundRobinPhaseSelectLogic vs Free         */
  int32_t visibility;            /* bins (10)                                     */
  int32_t sim_length;            /* 3600                                          */
  int32_t episode_limit;         /* 360                                           */
  int32_t initial_wait_time;     /* 0                                             */
  int32_t set_offset;            /* first flow set of replica 0                   */
  int32_t sequential_sets;       /* advance set on every restart                  */
  int32_t reward_kind;           /* 0 queue_length, 1 max_pressure                */
  double interval;               /* engine dt (1.0)                               */
  double veh_size_min_gap;       /* 7.5                                           */
  double flickering_coef;        /* 0.01                                          */
  int32_t kernel_family;         /* TSC_FAMILY_*: which kernel family steps the replicas */
  int32_t resident_capacity;     /* resident family: vehicle records per replica (0 = default) */
  int32_t resident_threads;      /* resident family: threads per workgroup: 768 (0 = default) */
  /* Runtime knobs of the resident family's data-dependent paths.  Results are the same, bit for bit, whatever they are
   * set to (tests/test_hip_paths.py runs every setting against the oracle); tsc_path_counts tells which path ran. */
  int32_t resident_force;        /* bit 0: every sub-step takes the two-phase hand-off path (else only when two hand-offs
                                  * touch the same list end); bit 1: the measure phase walks the vehicle lists (else only
                                  * after a vehicle left its list from behind another one) */
  int32_t resident_task_capacity;/* (vehicle, cross) tasks per sub-step held on chip, 1..768 (0 = 768); a vehicle whose
                                  * tasks do not fit scans its crosses serially */
  int32_t resident_margin;       /* AUTO handles: records of headroom below which a replica asks for the migration to
                                  * the list family.  0 = derived at create from the flow sets: the most vehicles any
                                  * set creates within the 16 env steps the host may run ahead of the flag, + 25 % for
                                  * the stage slots of the heavy vehicles among them, + 64 */
  int32_t resident_spill_limit;  /* AUTO handles: most replicas the spill pool may hold -- a replica that asks for more room is
                                  * promoted there one by one (TSC_FAMILY_AUTO above) -- before the whole handle moves to
                                  * the list family instead.  0 = max(16, n_replicas / 8); -1 = no pool: whole-handle
                                  * migration only (the behaviour of ABI 3) */
  int32_t replica_base;          /* global index of this handle's replica 0 in a job sharded over several handles / GPUs:
                                  * the counter-based draws (tsc_random_valid_actions, the observation noise) are keyed
                                  * on replica_base + r, so a replica draws the same numbers whatever the sharding */
  /* Observation noise on the device (reference common/observations.py:72-88, PositionMatrix._add_gaussian_noise): every
   * surviving entry (> 0) of an incoming lane's position-matrix window gets val + N(0, std), clipped to [0, 1], BEFORE the
   * integer truncation of the row (the entry ends as 1 iff val + noise >= 1); empty bins and stopped vehicles
   * vanish as without noise (Q2) and take no draw.  The draws are a counter-based generator keyed on (seed, replica, engine
   * step, signal, lane, ordinal of the surviving bin): the DRAWS are reproducible and independent across replicas -- the law
   * of the reference's noise, not numpy's stream (the single-env facade follows that one).  The bin value a draw is added
   * to is an f32 sum; on the resident family its terms are added with atomics in no fixed order, so an entry whose
   * val + noise lies within an ulp of 1 may differ between two runs (probability ~1e-7 per entry).  0 = off: rows bit for bit those of a
   * handle without the fields.  Both kernel families (ABI 5; the resident family draws where kr_step assembles the rows, the
   * list family in k_agent from per-bin values k_measure keeps: the same key gives the same draw on either).  A handle
   * with noise runs the full-size class of the resident family only and has no stand-by (no migration, no spill pool):
   * a replica that runs out of vehicle records there stalls with an error (tsc_errors). */
  double obs_noise_std;
  uint64_t obs_noise_seed;
} tsc_config_t;

/* Optional outputs of tsc_step; any pointer may be NULL (skipped). */
typedef struct tsc_step_out {
  float *obs_dev;            /* [R, A, 16*(9+vis)+20]  PositionMatrix obs          */
  float *state_dev;          /* [R, A, 16*12+20]       lane-feature state          */
  uint8_t *mask_dev;         /* [R, A, n_actions]      n_actions = 16|max n_phases or 2 */
  float *local_reward_dev;   /* [R, A]                                            */
  float *global_reward_dev;  /* [R]                                               */
  uint8_t *done_dev;         /* [R]  episode_over                                 */
  uint8_t *terminated_dev;   /* [R]  sim_step == sim_length                       */
  double *stats_dev;         /* [R, TSC_N_STATS]                                  */
  double *agent_stats_dev;   /* [R, A, TSC_N_AGENT_STATS]                         */
  int32_t *phase_dev;        /* [R, A, 3] current index, time_on_phase, phase_changed */
  double *local_reward64_dev;/* [R, A]  f64 copies for parity checks              */
  double *global_reward64_dev;/* [R] */
  uint8_t *valid_dev;        /* [R]  1: every output row of this replica is that of the step (or observation) just
                              * executed -- the reference never hands out anything else (pytsc/__init__.py:178-182);
                              * 0: the replica is STALLED (resident family, out of vehicle records: see TSC_FAMILY_AUTO
                              * above) -- its rows in the other buffers were not written by this call and still hold its
                              * last completed step.  Always 1 on the list family. */
  int16_t *phase_scores_dev; /* ABI 5 -- [R, A, TSC_MAX_PHASES, 3]: what the reference's rule-based controllers compute per
                              * phase from the position matrices (controllers/controllers.py): [0] stopped vehicles in the
                              * incoming windows of the phase's lanes (bins == 0.0; GreedyController :93-112), [1] vehicles
                              * in those windows (bins >= 0; SOTLController :227-242), [2] sum over the phase's incoming lanes
                              * of |vehicles in its window - vehicles in the outgoing window of its LAST outgoing lane|
                              * (MaxPressureController :153-178).  Phases beyond the signal's own: 0.  Written where the
                              * step assembles the observation rows (the window bits are on chip there): no second pass
                              * over the state, no [R, A, P, lanes] tensor on the caller's side. */
} tsc_step_out_t;

typedef struct tsc_env *tsc_handle_t;

int tsc_abi_version(void);

int tsc_create(const tsc_net_t *net, const tsc_flows_t *flows,
               const tsc_config_t *cfg, int device, tsc_handle_t *out);
int tsc_destroy(tsc_handle_t h);

/* Restart replicas with reset_mask_dev[r] != 0 (NULL = all): empties the
 * network, rewinds time, re-applies the start phase and runs
 * initial_wait_time warm-up sub-steps. */
int tsc_reset(tsc_handle_t h, const uint8_t *reset_mask_dev, void *stream);

/* tsc_reset with the flow set of each restarted replica chosen by the caller:
 * set_index_dev[R] int32, taken modulo n_sets; a negative entry (or a NULL
 * pointer) keeps tsc_reset's rule.  This is how `flow_rate_type: random`
 * (reference backends/cityflow/config.py:63-76: random.choice(flow_files) at
 * every simulated-hour restart) is served: the host side draws, as the reference
 * does, and hands the draw over. */
int tsc_reset_sets(tsc_handle_t h, const uint8_t *reset_mask_dev,
                   const int32_t *set_index_dev, void *stream);

/* Park the replicas with park_mask_dev[r] != 0: from now on tsc_step / tsc_observe skip them (their rows are not written,
 * valid_dev[r] = 0) until a tsc_reset / tsc_reset_sets that selects them starts them again from an empty network.  What
 * a batched multi-map environment needs (reference wrappers/epymarl.py:112-336 draws one of several maps at every
 * simulated-hour restart: with one handle per map, a slot's replicas in the maps it is not playing are parked and cost
 * a workgroup launch each).  Both kernel families (ABI 5).  On the resident family a parked replica keeps its state as it
 * is; on the list family its vehicle-level state is NOT kept consistent while it is parked (tsc_snapshot / tsc_lane_detail of
 * it are unspecified until the reset that starts it again) -- its rows are not written and valid_dev says so, which is all a
 * caller may rely on. */
int tsc_park(tsc_handle_t h, const uint8_t *park_mask_dev, void *stream);

/* One env step for all replicas: apply actions_dev[R, A], advance delta_time
 * engine sub-steps, measure, and fill the requested outputs.  An action of -1
 * leaves that signal untouched (phase and time_on_phase), as when the reference
 * does not call switch_phase for it; actions_dev == NULL leaves all untouched. */
int tsc_step(tsc_handle_t h, const int32_t *actions_dev,
             const tsc_step_out_t *out, void *stream);

/* Measure + fill outputs without stepping (state right after create/reset). */
int tsc_observe(tsc_handle_t h, const tsc_step_out_t *out, void *stream);

/* The reference's rule-based controllers (pytsc/controllers/controllers.py:26-268) for every replica and signal in ONE
 * launch, from the outputs of the step just made: phase_dev [R, A, 3] and phase_scores_dev [R, A, TSC_MAX_PHASES, 3] as
 * tsc_step wrote them.  actions_dev [R, A] int32 = the action to hand to the next tsc_step (a target phase index on the
 * phase_selection action space -- what controllers/evaluate.py:26,57-58 forces --, switch / stay on phase_switch).
 *   TSC_CTRL_FIXED_TIME   p0 = green time (25): stay while green and time_on_phase < p0, else the next phase (:39-54)
 *   TSC_CTRL_GREEDY       on a green phase the allowable phase with the most stopped vehicles, else the next (:69-112)
 *   TSC_CTRL_MAX_PRESSURE ... with the largest pressure (:127-178)
 *   TSC_CTRL_SOTL         p0 = theta (3), p1 = mu (4), p2 = phi_min (5) (:200-242)
 *   TSC_CTRL_RANDOM       uniform among the allowable phases (:245-268)
 * Allowable phases: TLSFreePhaseSelectLogic / TLSRoundRobinPhaseSelectLogic (common/traffic_signal.py:329-404) from
 * (phase, time on it).  Ties (the reference: np.random.choice among the maximisers) and the random rule draw from a
 * counter-based generator keyed on (seed, replica_base + replica, signal, calls so far): the same law, not numpy's
 * stream. */
#define TSC_CTRL_FIXED_TIME 0
#define TSC_CTRL_GREEDY 1
#define TSC_CTRL_MAX_PRESSURE 2
#define TSC_CTRL_SOTL 3
#define TSC_CTRL_RANDOM 4
int tsc_controller_actions(tsc_handle_t h, int32_t kind, int32_t p0, int32_t p1, int32_t p2, uint64_t seed,
                           const int32_t *phase_dev, const int16_t *phase_scores_dev, int32_t *actions_dev, void *stream);

/* Per-lane measurements of the last measure pass:
 *  lane_counts_dev [R, n_lanes, 2] (n_vehicles, n_queued) int32
 *  lane_speed_dev  [R, n_lanes]    sum of speeds (mean_speed = sum / n) double
 *  posmat_dev      [R, n_lanes, max_bins] full position matrices (double),
 *                  max_bins = max(max lane_bins, visibility); may be NULL. */
int tsc_lane_detail(tsc_handle_t h, int32_t *lane_counts_dev,
                    double *lane_speed_dev, double *posmat_dev,
                    int32_t max_bins, void *stream);

/* Per-lane counts over the visibility windows of the last measure pass, for the
 * rule-based controllers (reference controllers/controllers.py:93-242 read
 * position_matrix[-visibility:] of incoming and [:visibility] of outgoing lanes):
 *  window_counts_dev [R, n_lanes, 3] int32 =
 *    (# bins == 0.0 in the incoming window, # bins >= 0 in the incoming window,
 *     # bins >= 0 in the outgoing window). */
int tsc_lane_window_counts(tsc_handle_t h, int32_t *window_counts_dev, void *stream);

/* On-device random policy (reference controllers/controllers.py:245-268,
 * RandomController: a uniformly random allowable action per signal), fused into one
 * launch: for every (replica, agent) draws one of the actions whose entry in
 *  mask_dev      [R, A, n_actions] uint8 (as written by tsc_step / tsc_observe) is non-zero,
 * into
 *  actions_dev   [R, A] int32 (0 when no action is allowed).
 * The draw is a counter-based hash of (seed, number of calls on this handle so far,
 * replica, agent): reproducible for a seed, independent of the grid shape. */
int tsc_random_valid_actions(tsc_handle_t h, const uint8_t *mask_dev, uint64_t seed,
                             int32_t *actions_dev, void *stream);

/* Synchronous dump of one replica's vehicles to HOST arrays (capacity `cap`):
 * drivable index, distance, speed, vid.  Returns the vehicle count (>= 0) or a
 * negative error.  Vehicles inserted for the *next* sub-step are excluded. */
int tsc_snapshot(tsc_handle_t h, int32_t replica, int32_t cap, int32_t *drv,
                 double *dist, double *speed, int32_t *vid);

/* Synchronous: device-side error counters (slot overflow etc.), 0 = clean. */
int tsc_errors(tsc_handle_t h, int32_t *n_errors);

/* Synchronous: how often the resident family took each of its data-dependent paths since tsc_create,
 * counts[TSC_N_PATHS] int64 (all 0 for a handle on the list family):
 *  [0] (replica, sub-step)s whose hand-offs took the two-phase path (unlink per source list, ordered append per target)
 *  [1] (replica, env step)s whose measure phase walked the vehicle lists
 *  [2] (vehicle, sub-step)s that scanned their crosses serially because the on-chip task list was full
 *  [3] of [0]: not forced -- two hand-offs really touched the same list end in one sub-step
 *  [4] of [1]: not forced -- a vehicle really left its list from behind another one
 *  [5] replicas that ran out of records and stalled until the handle migrated
 *  [6] two-class handles: (replica, step)s the small class abandoned within the step and the full-size launch redid
 *  [7] two-class handles: class changes decided at the end of a step (either direction) */
#define TSC_N_PATHS 8
int tsc_path_counts(tsc_handle_t h, int64_t *counts);

/* The spill pool of an AUTO handle on the resident family (synchronous with the HOST's view only: no device call):
 * counts[4] int32 = (replicas living in the pool now, pool slots handed out since the last full restart, record capacity
 * of a pool slot, resident_spill_limit in force); all 0 for a handle without a pool. */
int tsc_spill_counts(tsc_handle_t h, int32_t *counts);

/* The classes of a two-class handle (resident family, ABI 5): replicas whose state needs few vehicle records are stepped by
 * 512-thread workgroups with an LDS image of at most 80 KB -- two of them share a CU, 1.5-1.7x the rate per replica --, the
 * others by the full-size workgroups; a replica changes class by a word of its blob's header, set by the kernel itself.
 * counts[2] int32 = (replicas in the small class, in the full-size class) now (synchronous); (0, R) without a small class. */
int tsc_class_counts(tsc_handle_t h, int32_t *counts);

/* Mean number of running vehicles per replica at the last measure pass
 * (synchronous; used for reporting V_a). */
int tsc_mean_vehicles(tsc_handle_t h, double *mean);

/* Vehicle bookkeeping of every replica: counts_dev [R, 3] int32 = (vehicles that have entered the
 * network since the last restart, vehicles running now, vehicles that have finished).  entered ==
 * running + finished is the conservation invariant (Engine.get_vehicle_count() is the second column,
 * reference retriever.py:108-112). */
int tsc_replica_counts(tsc_handle_t h, int32_t *counts_dev, void *stream);

/* Kernel timing with HIP events on the stream the kernels are launched on:
 * while enabled (on = n > 0), the launch of the dominant kernel (k_exec, one per
 * engine sub-step) is bracketed by an event pair on every n-th sub-step (an event
 * record costs a few microseconds of pipeline bubble; n = 1 brackets them all).
 * tsc_profile_read synchronises, returns the summed duration [ms] and the number
 * of bracketed launches since enable, and clears the accumulators. */
int tsc_profile_enable(tsc_handle_t h, int on);
int tsc_profile_read(tsc_handle_t h, double *total_ms, int32_t *n_launches);

/* Would tsc_create accept these tables, and with which kernel family?  Pure host code (no GPU needed): the
 * create-time requirements of the HIP path and, unless cfg->kernel_family is TSC_FAMILY_LIST, whether a
 * replica fits on chip.  Returns TSC_OK / TSC_EUNSUP / TSC_EINVAL; *family_out = the family tsc_create would pick,
 * *resident_capacity_out = vehicle records per replica of the resident family (0 for the list family). */
int tsc_check_tables(const tsc_net_t *net, const tsc_flows_t *flows, const tsc_config_t *cfg, int32_t *family_out,
                     int32_t *resident_capacity_out);

/* The kernel family the handle runs (TSC_FAMILY_LIST / TSC_FAMILY_RESIDENT). */
int tsc_kernel_family(tsc_handle_t h);

/* Which instantiation of the resident family's kernel the handle launches: 0 -- the generic one (layout and vehicle
 * constants from device tables), 1 -- layout-specialised (BASELINE configs 1, 3, 4 at their default capacity), 2 -- layout
 * AND a known vehicle as compile-time constants (the 4x4 grid with CityFlow's default vehicle -- the reference's trip
 * generators write no other: pytsc/backends/cityflow/trip_generator.py:47-57; Jinan with the vehicle of the shipped
 * real-world flow files).  Same results bit for bit; -1 for a handle on the list family. */
int tsc_kernel_variant(tsc_handle_t h);

/* Sizes derived from the tables. */
int tsc_obs_size(tsc_handle_t h);
int tsc_state_size(tsc_handle_t h);
int tsc_n_actions(tsc_handle_t h);

#ifdef __cplusplus
}
#endif
#endif /* PYTSC_AMD_H */
