/*
 * rdv_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, float64, one-env-at-a-time restatement of the reference algorithm for RendezvousEnv.step()/reset()
 * (cfdeinza/reinforcement-learning-rendezvous, rendezvous_env.py + utils/dynamics.py + utils/quaternions.py +
 * three functions of utils/general.py).  Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (librdv_hip.so) never links, loads or calls it, and has no CPU fallback.
 *
 * Pinning (see oracle/README.md): checked in tests/test_oracle_golden.py against
 *   (1) per-step transition tuples recorded from the unmodified reference env run in the build container
 *       (tests/golden/make_golden.py, tests/golden/steps_*.npz),
 *   (2) the reference's own published per-trajectory Monte Carlo table
 *       (results/data_monte_carlo_results_mlp.xlsx -> tests/golden/mc_published_xlsx.npz),
 *   (3) the known-answer cases of the reference's verification/ scripts,
 *   (4) direct calls of the reference's utils functions and of scipy's solve_ivp(RK45) on its right-hand side with random
 *       inertia tensors and torques (tests/golden/kat_reference_functions.npz, kat_rigid_body.npz), and reference
 *       transitions with anisotropic bodies (tests/golden/steps_F_rigid.npz).
 *
 * The one deliberate departure from the reference: the two scipy RK45 attitude integrations
 * (rendezvous_env.py:561-570, :588-597) are replaced by the exact solution of the same ODE for the
 * reference's isotropic inertia and zero torque (orc_integrate_attitude).  ORC_INTEGRATOR_RK45 keeps a
 * Dormand-Prince 5(4) integrator with scipy's step control for cross-checking that substitution, and
 * ORC_INTEGRATOR_GENERAL runs it on the full right-hand side (any inertia tensor, body torques: OrcRigidBody).
 */
#ifndef RDV_ORACLE_H_
#define RDV_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to RdvParams in include/rdv.h (57 doubles); declared separately on purpose so the
 * checker shares no code with the product.  tests/test_abi.py asserts the two sizes agree. */
typedef struct OrcParams {
  double nominal_rc0[3], nominal_vc0[3], nominal_qc0[4], nominal_wc0[3], nominal_qt0[4], nominal_wt0[3];
  double rc0_range, vc0_range, qc0_range, wc0_range, qt0_range, wt0_range;
  double dt, t_max;
  double max_delta_v, max_delta_w;
  double max_axial_distance, max_axial_speed, max_wc;
  double max_attitude_error;
  double koz_radius, corridor_half_angle;
  double corridor_axis[3], capture_axis[3], rd[3];
  double max_rd_error, max_vd_error, max_qd_error, max_wd_error;
  double bubble_radius0, bubble_decrease_rate, bubble_min;
  double n;
  double collision_coef, bonus_coef, fuel_coef, att_coef;
} OrcParams;

/* One environment: the attributes RendezvousEnv mutates (rendezvous_env.py:44-49, :68, :109-118). */
typedef struct OrcEnv {
  double rc[3], vc[3], qc[4], wc[3], qt[4], wt[3];   /* 20 state reals, CSV column order */
  double bubble_radius, total_delta_v, total_delta_w;
  double episode_return;   /* what SB3's Monitor would sum */
  int32_t k;               /* steps taken this episode; t = round(k*dt, 3) */
  int32_t collided;        /* latched */
  int32_t success;         /* count */
  int32_t episode;         /* resets performed so far (RNG / tape index of the NEXT reset) */
  int32_t halted;          /* ORC_ON_DONE_HALT: env finished and frozen */
} OrcEnv;

enum { ORC_STORAGE_F32 = 0, ORC_STORAGE_F64 = 1 };
enum { ORC_ON_DONE_RESET = 0, ORC_ON_DONE_HALT = 1, ORC_ON_DONE_NOTHING = 2 };
enum { ORC_INTEGRATOR_EXACT = 0, ORC_INTEGRATOR_RK45 = 1, ORC_INTEGRATOR_GENERAL = 2 };

/* The rigid-body attributes of the env (self.inertia / self.inv_inertia :75-80, self.inertia_target / self.inv_inertia_target
 * :96-101; row-major 3x3) and the torque arguments of integrate_chaser_attitude / integrate_target_attitude (:552, :579).
 * Used with ORC_INTEGRATOR_GENERAL: scipy's RK45 on the full right-hand side (dynamics.py:93-175). */
typedef struct OrcRigidBody {
  double inertia_chaser[9], inv_inertia_chaser[9], torque_chaser[3];
  double inertia_target[9], inv_inertia_target[9], torque_target[3];
  double rtol, atol;      /* :567-568: 1e-7, 1e-6 */
} OrcRigidBody;

typedef struct OrcConfig {
  int32_t storage;        /* ORC_STORAGE_F32 rounds the state/aux to float after every update, as the HIP
                             production path stores it; ORC_STORAGE_F64 is the reference's own precision */
  int32_t on_done;
  int32_t integrator;
  int32_t tape_depth;     /* 0 = Philox resets */
  int32_t numpy_legacy;   /* 0: NumPy >= 2 (NEP 50) scalar promotion, what the reference does when run today and what
                             the golden vectors were recorded under; 1: NumPy 1.23.3 promotion (the version pinned by
                             the checkpoint's system_info.txt).  See the header of rdv_oracle.c. */
  int32_t reserved;
  const double* tape;     /* [depth][n][20] */
  uint64_t seed;
  uint64_t env_id_offset;
  const OrcRigidBody* rigid;   /* ORC_INTEGRATOR_GENERAL only */
} OrcConfig;

typedef struct OrcStats {
  uint64_t env_steps, episodes, successes, collisions, reasons[4];
  double sum_return, sum_length, sum_delta_v, sum_delta_w;
} OrcStats;

typedef struct OrcStepOut {
  float* obs; double* reward; uint8_t* done;
  float* terminal_obs; double* episode_return; int32_t* episode_length; uint8_t* done_reason; double* diag;
} OrcStepOut;

int  orc_version(void);
void orc_params_default(OrcParams* p);

/* 24 uniforms in (0,1) for (seed, global env id, episode): Philox4x32-10, 4 blocks, 21 bits per uniform. */
void orc_philox_uniforms(uint64_t seed, uint64_t env_id, uint32_t episode, double u[24]);
void orc_philox_block(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t block, uint32_t out[4]);

/* scalar pieces (exposed so tests can check them one by one against the importable reference functions) */
void   orc_quat2mat(const double q[4], double m[9]);                               /* quaternions.py:48-68 */
void   orc_rot2quat(const double axis[3], double theta, double q[4]);              /* quaternions.py:11-27 */
void   orc_quat_product(const double q1[4], const double q2[4], double out[4]);    /* quaternions.py:149-170 */
void   orc_cw_solution(const double r0[3], const double v0[3], double n, double t, double r[3], double v[3]); /* dynamics.py:24-55 */
double orc_angle_between(const double a[3], const double b[3]);                     /* general.py:163-181 */
void   orc_integrate_attitude(double q[4], double w[3], double dt, int integrator); /* rendezvous_env.py:552-604 */
void   orc_att_rhs(const double y[7], double dy[7]);                                /* dynamics.py:93-175 (I = 16.67*1, torque 0) */
void   orc_att_rhs_general(const double y[7], const double inertia[9], const double inv_inertia[9], const double torque[3],
                           double dy[7]);                                            /* dynamics.py:93-175, any tensor / torque */
int    orc_solve_attitude_rk45(double y[7], double dt, const double inertia[9], const double inv_inertia[9],
                               const double torque[3], double rtol, double atol);    /* the solve_ivp call of :561-570; returns nfev */
void   orc_integrate_attitude_general(double q[4], double w[3], double dt, const double inertia[9], const double inv_inertia[9],
                                      const double torque[3], double rtol, double atol);   /* :552-604 incl. the normalisation */

/* env-level pieces */
void   orc_get_observation(const OrcParams* p, const OrcEnv* e, float obs[17]);    /* :294-311 */
void   orc_get_errors(const OrcParams* p, const OrcEnv* e, double err[4]);         /* :451-468 */
int    orc_check_collision(const OrcParams* p, const OrcEnv* e);                   /* :388-404 */
int    orc_check_success(const OrcParams* p, const OrcEnv* e);                     /* :406-422 */
double orc_dist_from_koz(const OrcParams* p, const OrcEnv* e);                     /* :510-537 */
void   orc_diagnose(const OrcParams* p, const OrcEnv* e, double diag[8]);

/* batch API used by the tests and by bench.py's cpu_baseline leg.  n_threads > 1 uses OpenMP over envs. */
void orc_reset(const OrcParams* p, const OrcConfig* c, int64_t n, OrcEnv* envs, const uint8_t* mask, float* obs_out);
void orc_step(const OrcParams* p, const OrcConfig* c, int64_t n, OrcEnv* envs, const float* actions,
              const OrcStepOut* out, OrcStats* stats, int n_threads);
void orc_set_state(int64_t n, OrcEnv* envs, const double* states, int storage);
void orc_get_state(int64_t n, const OrcEnv* envs, double* states);
void orc_get_aux(const OrcParams* p, int64_t n, const OrcEnv* envs, double* aux);
void orc_observe(const OrcParams* p, int64_t n, const OrcEnv* envs, float* obs);
void orc_diagnose_batch(const OrcParams* p, int64_t n, const OrcEnv* envs, double* diag);
int64_t orc_sizeof_env(void);
int64_t orc_sizeof_params(void);

#ifdef __cplusplus
}
#endif
#endif
