/*
 * chem_mi355.h -- C ABI of libchem_mi355.so, the MI355X-native reactive-MD inner loop
 * that replaces the modified-ESPResSo++ back end ChemLab drives.
 *
 * The reference (cgchemlab/chemlab) has no C ABI: its hot path is reached through
 * Boost.Python `espressopp.*` objects.  Every entry point below therefore cites the
 * ChemLab call site (file:line under /root/reference) whose espressopp call it replaces.
 * INTEGRATION.md shows the py3 `espressopp`-shaped shim (ctypes) a maintainer binds
 * on top of these symbols.
 *
 * Conventions
 *  - every function returns 0 (or a count >= 0) on success and a negative CHEM_E* code on
 *    failure; the message is available from chem_last_error(ctx);
 *  - the caller owns every input buffer; it is copied during the call;
 *  - the context owns all device memory; outputs go to caller-allocated buffers with a
 *    capacity, the return value is the element count (CHEM_ENOSPC if cap is too small);
 *  - a context is bound to one GPU and is not thread-safe; multi-GPU = one context per
 *    rank (one process per GPU), joined with chem_comm_init();
 *  - there is NO CPU fall-back behind these symbols: without a gfx950 device
 *    chem_create() fails.
 */
#ifndef CHEM_MI355_H
#define CHEM_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHEM_ABI_VERSION 1

/* error codes */
#define CHEM_OK        0
#define CHEM_EINVAL   -1   /* bad argument */
#define CHEM_ENOSPC   -2   /* output capacity too small / internal capacity exceeded */
#define CHEM_EDEVICE  -3   /* HIP error or no usable device */
#define CHEM_ESTATE   -4   /* call sequence error (e.g. run before particles) */
#define CHEM_ENOTIMPL -5   /* feature outside the hot-path scope */
#define CHEM_ECOMM    -6   /* RCCL error */

/* arithmetic precision of the device path (chem_create) */
#define CHEM_PREC_F32 32   /* fp32 positions/velocities/forces, fp64 bonded + reaction distance */
#define CHEM_PREC_F64 64   /* everything fp64 (parity mode)                                    */

/* bonded potential kinds (chem_list_create).  Parameter vectors `p` for chem_list_set_params:
 *   HARMONIC        K, r0                 U = K (r-r0)^2            gromacs_topology.py:918,949-961
 *   FENE            K, r0, rMax           U = -K/2 rMax^2 ln(1-((r-r0)/rMax)^2)      :926-931
 *   ANG_HARMONIC    K, theta0[rad]        U = K (theta-theta0)^2                    :1073,1086
 *   ANG_COSINE      K, theta0[rad]        U = K (1+cos(theta-theta0))               :1082
 *   DIH_NCOS        K, phi0[rad], n       U = K (1+cos(n phi - phi0))               :1185-1190
 *   DIH_RB          C0..C5                U = sum_n C_n cos^n(phi-pi)               :1192-1198
 */
#define CHEM_POT_HARMONIC      1
#define CHEM_POT_FENE          2
#define CHEM_POT_TABULATED     3   /* bonds: params = { table handle } from chem_table_create */
#define CHEM_POT_FENE_LJ       4   /* bonds func 9, FENELennardJones(K, r0, rMax, sigma, epsilon): FENE + 4 eps ((s/r)^12 - (s/r)^6),
                                      doc/topology.rst:68-75, gromacs_topology.py:935-942 */
#define CHEM_POT_LJ_BOND       5   /* 1-4 pairs, FixedPairListLennardJones(epsilon, sigma, cutoff): plain LJ inside the cutoff,
                                      gromacs_topology.py:1314-1411 */
#define CHEM_POT_ANG_HARMONIC 10
#define CHEM_POT_ANG_COSINE   11
#define CHEM_POT_ANG_TABULATED 12  /* angles func 8: params = { table handle }, grid in radians, f = -dU/dtheta */
#define CHEM_POT_DIH_NCOS     20
#define CHEM_POT_DIH_RB       21
#define CHEM_POT_DIH_HARMONIC  23  /* dihedrals func 12, DihedralHarmonic(K, phi0): U = K/2 (phi - phi0)^2, difference wrapped to
                                      (-pi, pi], doc/topology.rst:120-128, gromacs_topology.py:1199-1202 */
#define CHEM_POT_DIH_TABULATED 22  /* dihedrals func 8: params = { table handle }, grid in radians over [-pi, pi], f = -dU/dphi */
#define CHEM_MAX_POT_PARAMS    6
#define CHEM_MAX_LISTS        32
#define CHEM_MAX_TYPES        16
#define CHEM_MAX_REACTIONS    16

/* selectors for chem_get_state; all outputs are in ascending particle-id order */
#define CHEM_STATE_POS     1   /* double[n*3], folded into the box                 */
#define CHEM_STATE_VEL     2   /* double[n*3]                                      */
#define CHEM_STATE_FORCE   3   /* double[n*3], forces of the last evaluation       */
#define CHEM_STATE_TYPE    4   /* int32[n]                                         */
#define CHEM_STATE_STATE   5   /* int32[n]  chemical state                         */
#define CHEM_STATE_RESID   6   /* int32[n]                                         */
#define CHEM_STATE_MASS    7   /* double[n]                                        */
#define CHEM_STATE_ID      8   /* int64[n]                                         */
#define CHEM_STATE_IMAGE   9   /* int32[n*3] periodic image counters               */
#define CHEM_STATE_MOLID  10   /* int32[n]  lowest particle id of the bonded cluster */
#define CHEM_STATE_POS_UNFOLDED 11 /* double[n*3] = pos + image*L                  */

typedef struct chem_ctx chem_ctx;

/* One chemical reaction  T1(min1,max1) + T2(min2,max2) -> N1(d1):N2(d2).
 * Replaces espressopp.integrator.Reaction(type_1,type_2,delta_1,delta_2,min_state_1,
 * max_state_1,min_state_2,max_state_2,rate,fpl,cutoff) + .intramolecular/.intraresidual/
 * .is_virtual/.active, get_reaction_cutoff().min_cutoff and the PostProcessChangeProperty
 * type change (reaction_setup.py:81-163). */
typedef struct chem_reaction_desc {
  int32_t type_1, type_2;
  int32_t delta_1, delta_2;
  int32_t min_state_1, max_state_1;   /* half-open [min,max) */
  int32_t min_state_2, max_state_2;
  double  rate;
  double  cutoff;
  double  min_cutoff;
  int32_t intramolecular;             /* 1: partners may belong to one bonded cluster */
  int32_t intraresidual;              /* 1: partners may share res_id                 */
  int32_t is_virtual;                 /* 1: no bond is created                        */
  int32_t active;
  int32_t bond_list;                  /* handle of the arity-2 list receiving new bonds */
  int32_t new_type_1, new_type_2;     /* -1: unchanged                                */
  double  new_mass_1, new_mass_2;     /* used when the type changes                   */
  double  new_q_1, new_q_2;
} chem_reaction_desc;

/* One accepted reaction event; chem_get_events returns them sorted by
 * (step, min(id_a,id_b), max(id_a,id_b)). */
typedef struct chem_event {
  int64_t step;
  int64_t id_a, id_b;     /* particle taking role type_1 / type_2 */
  int32_t reaction;
  int32_t pad;            /* with option "count_intra_inter": 1 = both particles were in one bonded cluster before the event */
  double  r2;             /* fp64 squared distance at decision time */
} chem_event;

typedef struct chem_obs {
  int64_t step;
  int64_t npart;
  double  ekin;
  double  temperature;             /* 2 Ekin / (3 N), in energy units (k_B T)  */
  double  epot_lj;                 /* VerletListLennardJones                   */
  double  epot_tab;                /* VerletListTabulated                      */
  double  epot_list[CHEM_MAX_LISTS];
  int64_t list_size[CHEM_MAX_LISTS];
  double  momentum[3];
  double  virial_nb;               /* sum_pairs r.F of non-bonded              */
} chem_obs;

typedef struct chem_timers {
  double run_wall_s;               /* wall seconds inside chem_run (== integratorLoop, start_simulation.py:779-781) */
  int64_t steps;
  int64_t rebuilds;
  int64_t reaction_steps;
  int64_t nlist_entries;           /* entries of the current full neighbour list */
  int64_t nlist_capacity;
  double  reaction_wall_s;
  double  rebuild_wall_s;          /* host-observed; only reaction-step/forced rebuilds are synchronous */
  double  pair_kernel_ms;          /* HIP-event time of all pair-force launches of the last chem_run */
  int64_t pair_kernel_launches;
  /* HIP-event samples of the per-step neighbour kernel (k_rebuild_fused) of the last chem_run, split
   * into launches that rebuilt the lists and launches that only took the decision */
  double  rebuild_kernel_ms;
  int64_t rebuild_kernel_launches;
  double  decide_kernel_ms;
  int64_t decide_kernel_launches;
  int64_t nlist_entries_all;       /* pairs within rc+skin before the type-pair filter (0 if unknown) */
  /* HIP-event samples of the remaining per-step kernels of the last chem_run */
  double  integrate_kernel_ms;
  int64_t integrate_kernel_launches;
  double  bonded_kernel_ms;
  int64_t bonded_kernel_launches;
  /* list builds actually performed (`rebuilds` above counts the reference rule on the workload's skin; with a wider internal
   * list skin -- option list_skin -- the lists are rebuilt less often) */
  int64_t list_rebuilds;
} chem_timers;

/* ---- life cycle ---------------------------------------------------------------------- */
/* espressopp.System() + esutil.RNG + storage (start_simulation.py:148-163) */
chem_ctx*   chem_create(int device_id, int precision);
void        chem_destroy(chem_ctx* ctx);
const char* chem_last_error(chem_ctx* ctx);      /* ctx may be NULL: last create error */
int         chem_abi_version(void);

/* ---- system set-up ------------------------------------------------------------------- */
/* bc.OrthorhombicBC(rng, box)  start_simulation.py:162 */
int chem_set_box(chem_ctx* ctx, const double L[3]);
/* VerletList(system, cutoff=max_cutoff, ...) + system.skin  start_simulation.py:151,193-197 */
int chem_set_cutoff(chem_ctx* ctx, double max_cutoff, double skin);
/* integrator.dt = dt  start_simulation.py:166 */
int chem_set_dt(chem_ctx* ctx, double dt);
/* storage.addParticles(particle_list,'id','type','pos','mass','q','res_id','state',..)
 * + decompose()  start_simulation.py:169-171, gromacs_topology.py:1418-1441.
 * vel may be NULL (zeros); q, state, res_id may be NULL (0 / 0 / id). */
int chem_set_particles(chem_ctx* ctx, int64_t n, const int64_t* id, const int32_t* type,
                       const double* pos, const double* vel, const double* mass,
                       const double* q, const int32_t* state, const int32_t* res_id);
/* storage.modifyParticle(pid, 'type'|'state'|'mass', v)  examples/atrp_lj/hooks.py:63-65 */
int chem_modify_particle(chem_ctx* ctx, int64_t id, int what /*CHEM_STATE_TYPE|STATE|MASS|RESID*/, double value);
/* DynamicExcludeList(integrator, gt.exclusions)  start_simulation.py:189 */
int chem_set_exclusions(chem_ctx* ctx, int64_t n, const int64_t* id_pairs);

/* ---- non-bonded ---------------------------------------------------------------------- */
/* interaction.LennardJones(epsilon, sigma, cutoff) via VerletListLennardJones.setPotential
 * gromacs_topology.py:715-721.  shift_auto!=0: U(rc)=0. */
int chem_nb_lj(chem_ctx* ctx, int t1, int t2, double eps, double sig, double rc, int shift_auto);
/* interaction.Tabulated(itype=1, filename, cutoff) via VerletListTabulated.setPotential
 * gromacs_topology.py:696-707; rows are the `r e f` lines of the .pot file
 * (tools/convert_gromacs2espp.py:84-107), uniform spacing dr starting at r0. */
int chem_nb_table(chem_ctx* ctx, int t1, int t2, int64_t nrow, double r0, double dr,
                  const double* e, const double* f, double rc);

/* ---- bonded lists -------------------------------------------------------------------- */
/* FixedPairList/TripleList/QuadrupleList(storage) + FixedXListYyy(system, list, potential)
 * gromacs_topology.py:949-961,1086-1096,1206-1224; reaction_setup.py:444-467.
 * by_types!=0 -> the "Types" variant: parameters selected per entry from the CURRENT
 * particle types (gromacs_topology.py:969-981).  Returns the list handle. */
int chem_list_create(chem_ctx* ctx, int arity, int potential_kind, int by_types);
/* interaction.Tabulated(itype=1, filename=table_b<N>.pot) for bonds, `[ bonds ]`/`[ bondtypes ]` func 8
 * gromacs_topology.py:919-925,953,961: rows `r e f` on a uniform grid (tools/convert_gromacs2espp.py), linear
 * interpolation of e(r) and f(r), F_ij = f(r)/r * r_ij, first/last row beyond the grid.  Returns a table
 * handle >= 0 that is passed as the single parameter of a CHEM_POT_TABULATED list (plain or per type pair).
 * The same registry serves interaction.TabulatedAngular (angles func 8, table_a<N>.pot, gromacs_topology.py:1074-1080):
 * grid in radians, columns U(theta) and -dU/dtheta, list kind CHEM_POT_ANG_TABULATED; and interaction.TabulatedDihedral
 * (dihedrals func 8, table_d<N>.pot, gromacs_topology.py:1192-1198): grid over [-pi, pi], kind CHEM_POT_DIH_TABULATED. */
int chem_table_create(chem_ctx* ctx, int64_t nrow, double r0, double dr, const double* e, const double* f);
/* addBonds/addTriples/addQuadruples: ids is n*arity particle ids */
int chem_list_add(chem_ctx* ctx, int list, int64_t n, const int64_t* ids);
/* plain list: t1..t4 ignored (pass -1); Types list: setPotential(type1,type2[,type3[,type4]],pot) */
int chem_list_set_params(chem_ctx* ctx, int list, int t1, int t2, int t3, int t4,
                         const double* p, int np);
/* getAllBonds/getAllTriples/getAllQuadruples: out receives count*arity ids */
int64_t chem_get_list(chem_ctx* ctx, int list, int64_t* out, int64_t cap_entries);

/* ---- thermostat ---------------------------------------------------------------------- */
/* integrator.LangevinThermostat: .temperature (=T*kb), .gamma  start_simulation.py:330-336.
 * gamma<=0 or kT<0 switches it off (thermostat=no, Q6 in SURVEY). */
int chem_thermostat_langevin(chem_ctx* ctx, double kT, double gamma, uint64_t seed);
/* LangevinThermostat.add_valid_types(thermal_groups)  start_simulation.py:312-336: friction and noise act only on
 * particles whose CURRENT type is listed (types change through reactions); n = 0 restores "every type" (the
 * reference passes an empty list when neither --thermal_groups nor --table_groups is given). */
int chem_thermostat_langevin_types(chem_ctx* ctx, int n, const int32_t* types);
/* integrator.BerendsenThermostat(system) (.temperature, .tau) and integrator.Isokinetic(system) (.temperature,
 * .coupling) -- start_simulation.py:341-348 (`thermostat = br | iso`): velocity rescaling after the second half
 * kick of a step (aftIntV).  With kT_now = 2 Ekin / (3 N):
 *   kind 1 Berendsen : v *= sqrt(1 + dt/tau * (kT/kT_now - 1)) every step            (param = tau)
 *   kind 2 Isokinetic: v *= sqrt(kT/kT_now) every `param` steps                       (param = coupling, >= 1)
 *   kind 0 switches it off.  Independent of chem_thermostat_langevin (the driver uses one of them). */
int chem_thermostat_rescale(chem_ctx* ctx, int kind, double kT, double param);
/* integrator.StochasticVelocityRescaling(system) (.temperature, .coupling) -- start_simulation.py:337-340
 * (`thermostat = vr`): every step after the second half kick, v *= sqrt(K_new / K) with K_new drawn from the
 * canonical kinetic-energy distribution relaxing with time constant `coupling` (Bussi-Donadio-Parrinello 2007;
 * 3 N degrees of freedom; stream keyed (seed, step), include/chem_philox.h svr_lambda).  coupling <= 0 switches it
 * off.  Shares the slot of chem_thermostat_rescale (the last call of either wins). */
int chem_thermostat_svr(chem_ctx* ctx, double kT, double coupling, uint64_t seed);
/* integrator.CapForce(system, max_force), added before the thermostat -- start_simulation.py:320-324
 * (`max_force`, app_args default -1 = off): after every force evaluation the conservative force of a
 * particle is rescaled to |f| = max_force where it exceeds it; the thermostat's friction and noise come
 * on top, uncapped (extension order).  max_force <= 0 switches it off. */
int chem_cap_force(chem_ctx* ctx, double max_force);

/* ---- reactions ----------------------------------------------------------------------- */
/* integrator.ChemicalReaction(system, vl, storage, tm, interval) + .nearest_mode
 * + .max_per_interval  reaction_setup.py:416-427 */
int chem_reaction_init(chem_ctx* ctx, int interval, int nearest, int max_per_interval, uint64_t seed);
/* ar.add_reaction(Reaction(...))  reaction_setup.py:81-92,506; returns the reaction index */
int chem_reaction_add(chem_ctx* ctx, const chem_reaction_desc* d);
/* topology_manager.register_tuple/triplet/quadruplet(list, t1, t2[, t3[, t4]])
 * start_simulation.py:395-440: new bonds spawn entries of `list` when the type tuple matches */
int chem_topology_register(chem_ctx* ctx, int arity, int list, const int32_t* types);
/* integrator.PostProcessChangeNeighboursProperty(tm).add_change_property(old_type, TopologyParticleProperties, nb_level),
 * attached to the reactions of a group by `extensions=` (reaction_post_process.py:76-115, reaction_setup.py:128-165;
 * examples/atrp_lj/atrp.cfg `type_transfers=MA:2->PA,ML:1->PL(state=1)`).  After the events of a reaction step have
 * been applied and the new bonds are in the graph, every particle exactly `nb_level` bonds away from a reactant of an
 * event of `reaction` (invoke_on 1: the type_1 role, 2: the type_2 role, 3: both) whose type is `old_type` gets
 * new_type / new_mass / new_q and, if set_state != 0, chemical state new_state.  Events are visited in canonical order
 * (min id, max id), role 1 before role 2, rules in the order they were added. */
typedef struct chem_nb_change {
  int32_t reaction, invoke_on, old_type, nb_level;
  int32_t new_type, set_state, new_state, pad;   /* set_state 0: keep the state, 1: state = new_state, 2: state += new_state (incr_state) */
  double  new_mass, new_q;
  int32_t min_state, max_state;                  /* only neighbours whose state is in [min, max) change (set_min_max_state); min >= max: no window */
} chem_nb_change;
int chem_reaction_neighbour_change(chem_ctx* ctx, const chem_nb_change* rule);
/* reaction.add_constraint(ReactionConstraintNeighbourState(type, min_state, max_state), 'type_1' | 'type_2')
 * reaction_setup.py:203-204 (exchange reactions `A:B + C -> A:C + B`, which the reference sets up as a virtual A + C reaction
 * that requires A to carry a B): a candidate pair is accepted only if the particle in role `role` (1 | 2) has a bonded
 * neighbour of type nb_type whose chemical state is in [min_state, max_state).  One constraint per reaction and role. */
int chem_reaction_constraint(chem_ctx* ctx, int reaction, int role, int nb_type, int min_state, int max_state);
/* integrator.RestrictReaction(...).define_connection(b1, b2)  reaction_setup.py:75-78,115-128 (group option
 * `connectivity_map`: a file of id pairs): reaction `reaction` only accepts candidate pairs whose unordered id pair was
 * defined; everything else about the reaction is unchanged.  Calls accumulate. */
int chem_reaction_restrict(chem_ctx* ctx, int reaction, int64_t n, const int64_t* id_pairs);
/* integrator.ATRPActivator(system, interval, num_particles, ratio_activator, ratio_deactivator, delta_catalyst,
 *     k_activate, k_deactivate) + .select_from_all + add_reactive_center(type_id, state, is_activator, new_property,
 *     delta_state)  -- src/chemlab/reaction_post_process.py:380-426, examples/atrp_lj/atrp.cfg:15-25.  An integrator
 * extension that fires every `interval` steps (after the reaction step when both fall on the same step -- the driver
 * adds it behind `ar`, start_simulation.py:737-740) and flips reactive centres between dormant and active:
 *   pool      = every particle (select_from_all != 0) or every particle matching a centre's (type, state);
 *   selection = the num_particles members of the pool with the smallest key (Philox stream keyed (seed, step, tag):
 *               out[0] = key, ties by tag; out[1] = acceptance uniform); visited in that order;
 *   a selected particle matching centre c (first match in insertion order) is flipped with probability
 *               k_activate * ratio_activator   when c.is_activator == 0  ("A" centres: consume activator),
 *               k_deactivate * ratio_deactivator when c.is_activator != 0 ("DA" centres: consume deactivator);
 *   a flip sets type/mass/q to the centre's new property, state += delta_state, and moves
 *               delta_catalyst / num_particles of catalyst from the consumed pool to the other one (clamped at 0).
 * The arithmetic lives in the external ESPResSo++ fork; this rule set is this build's restatement of the call
 * site's contract ([EXT-RECALL]).  desc == NULL disconnects the extension. */
typedef struct chem_atrp_desc {
  int32_t interval, num_particles, select_from_all, pad;
  double  ratio_activator, ratio_deactivator, delta_catalyst, k_activate, k_deactivate;
  uint64_t seed;
} chem_atrp_desc;
typedef struct chem_atrp_stats {     /* one row per firing (the reference writes them to `stats_file`) */
  int64_t step, candidates, selected, activated, deactivated;
  double  ratio_activator, ratio_deactivator;
} chem_atrp_stats;
int chem_atrp_init(chem_ctx* ctx, const chem_atrp_desc* desc);
int chem_atrp_add_center(chem_ctx* ctx, int type, int state, int is_activator, int new_type, double new_mass, double new_q, int delta_state);
int64_t chem_atrp_get_stats(chem_ctx* ctx, chem_atrp_stats* out, int64_t cap);
/* integrator.addExtension(ar) / ar.disconnect()  start_simulation.py:735-741,776-777 */
int chem_reactions_enable(chem_ctx* ctx, int on);
/* per-reaction rate update (Arrhenius hook, start_simulation.py:785-796) */
int chem_reaction_set_rate(chem_ctx* ctx, int reaction, double rate);

/* ---- the hot call -------------------------------------------------------------------- */
/* integrator.run(n)  start_simulation.py:780.  No host round trip per step. */
int chem_run(chem_ctx* ctx, int64_t nsteps);

/* ---- read-back ----------------------------------------------------------------------- */
int64_t chem_num_particles(chem_ctx* ctx);
int64_t chem_get_step(chem_ctx* ctx);
/* storage.getParticle(pid).pos/.v/.type/...; returns number of particles written */
int64_t chem_get_state(chem_ctx* ctx, int what, void* out, int64_t cap_elems);
int64_t chem_get_events(chem_ctx* ctx, chem_event* out, int64_t cap);
int64_t chem_get_exclusions(chem_ctx* ctx, int64_t* out_pairs, int64_t cap_pairs);
/* current Verlet list as unique id pairs (id_lo,id_hi), sorted; for tests/diagnostics */
int64_t chem_get_verlet_pairs(chem_ctx* ctx, int64_t* out_pairs, int64_t cap_pairs);
/* analysis.Temperature/KineticEnergy/PotentialEnergy/NFixedPairListEntries  start_simulation.py:453-493 */
int chem_observe(chem_ctx* ctx, chem_obs* out);
/* integrator.getTimers()/verletlist.get_timers()  start_simulation.py:1040-1076 */
int chem_get_timers(chem_ctx* ctx, chem_timers* out);
int chem_device_sync(chem_ctx* ctx);

/* ---- tuning -------------------------------------------------------------------------- */
/* neighbours per particle the list can hold (0 = automatic from density) */
int chem_set_nlist_capacity(chem_ctx* ctx, int max_neighbours);
/* generic integer knobs, see DESIGN.md ("tpp": lanes per particle in the pair kernel, ...) */
int chem_set_option(chem_ctx* ctx, const char* name, double value);

/* ---- multi-GPU (spatial domain decomposition, RCCL halo) ------------------------------ */
/* storage.DomainDecomposition(system, nodeGrid, cellGrid) under mpirun
 * start_simulation.py:152-163.  uid comes from chem_comm_unique_id on rank 0. */
int chem_comm_unique_id(char uid[128]);
int chem_comm_init(chem_ctx* ctx, int nranks, int rank, const int node_grid[3], const char uid[128]);
/* Same decomposition with the ranks living in ONE process (one host thread per context) and
 * exchanging by device-to-device copies through a hub named `hub_id`: several domains on a single
 * GPU; used to validate the multi-rank path without a multi-GPU node. */
int chem_comm_init_local(chem_ctx* ctx, int nranks, int rank, int hub_id);
/* Same decomposition with one PROCESS per rank and the ghost/migration buffers mapped through hipIpc memory handles
 * (rendezvous in the POSIX shared-memory segment `shm_name`, the same fresh name on every rank): ranks may share a
 * device, which RCCL refuses -- the multi-process flow of `bench.py --gpus N` on a box with fewer than N GPUs. */
int chem_comm_init_ipc(chem_ctx* ctx, int nranks, int rank, const char* shm_name);

#ifdef __cplusplus
}
#endif
#endif /* CHEM_MI355_H */
