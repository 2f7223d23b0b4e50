/*
 * rdv_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rdv_oracle.h for scope and pinning.
 *
 * CPU restatement (float64, scalar, one env at a time) of the reference hot path.  Citations are
 * file:line in cfdeinza/reinforcement-learning-rendezvous.
 *
 * NumPy dtype semantics restated on purpose (actions arrive as float32, as SB3 and monte_carlo.py pass them).
 * They depend on the NumPy version the reference runs under; OrcConfig.numpy_legacy selects the set:
 *
 *   expression (rendezvous_env.py)                                  NumPy >= 2 (NEP 50)      NumPy 1.23.3 (legacy)
 *   a[0:3] * max_delta_v      f32 array * Python float   :172       float32 product          float32 product
 *   a[3:]  * max_delta_w      f32 array * np.float64     :173       float64 product          float32 product
 *   abs(a[0:3]).sum() * max_delta_v   f32 * Python float :201       float32, f32 accumulate  float64
 *   abs(a[3:]).sum()  * max_delta_w   f32 * np.float64   :202       float64                  float64
 *   dt*fuel_coef * abs(a[0:3]).sum() / (3*max_delta_v)   :333       float32 arithmetic       float64
 *
 * (max_delta_w is an np.float64 because it is derived from inertia[0,0], :82; max_delta_v is a Python float, :81.)
 * The golden step vectors were recorded in the build container under NumPy 2.2, i.e. the NEP 50 column, which is
 * the default here and what the HIP product implements.  The two columns differ by float32 rounding of quantities
 * of order 1e-2 (<= 4e-10 per step in wc, <= 1e-6 in total_delta_v); tests/test_oracle_golden.py shows the
 * Monte Carlo outcome table is identical under both.
 */
#include "rdv_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_version(void) { return 1; }
int64_t orc_sizeof_env(void) { return (int64_t)sizeof(OrcEnv); }
int64_t orc_sizeof_params(void) { return (int64_t)sizeof(OrcParams); }

static const double ORC_PI = 3.14159265358979323846;

static double radians(double deg) { return deg * (ORC_PI / 180.0); } /* np.radians */
/* np.linalg.norm(v) of a 1-D float64 array is sqrt(v.dot(v)), and v.dot(v) / np.dot(a, b) go to the BLAS ddot: with the NumPy the golden
 * vectors were recorded under (2.2.6, bundled OpenBLAS) that is the ascending chain of fused multiply-adds below for 3 and 4
 * elements — verified on 20,000 random vectors each, bit for bit, and pinned by tests/golden/boundary_diag.npz, where a norm sits
 * 0-2 ulp from a limit.  (A BLAS without FMA sums rounded products: the comparisons then fall differently in ~1 of 10^15 states.) */
static double norm3(const double v[3]) { return sqrt(fma(v[2], v[2], fma(v[1], v[1], v[0] * v[0]))); }
static double norm4(const double q[4]) { return sqrt(fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0])))); }
static double dot3(const double a[3], const double b[3]) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void matvec(const double m[9], const double v[3], double out[3]) { /* np.matmul(M, v) */
  for (int i = 0; i < 3; ++i) out[i] = m[3 * i] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
}
static void matTvec(const double m[9], const double v[3], double out[3]) { /* np.matmul(M.T, v) */
  for (int i = 0; i < 3; ++i) out[i] = m[i] * v[0] + m[3 + i] * v[1] + m[6 + i] * v[2];
}
static double canon(double x, int storage) { return storage == ORC_STORAGE_F32 ? (double)(float)x : x; }

/* ---------------------------------------------------------------- defaults: rendezvous_env.py:52-126, :313 */
void orc_params_default(OrcParams* p) {
  memset(p, 0, sizeof(*p));
  p->nominal_rc0[1] = -10.0;                 /* :52 */
  p->nominal_qc0[0] = 1.0;                   /* :54 */
  p->nominal_qt0[0] = 1.0;                   /* :56 */
  p->rc0_range = 1.0;                        /* :60 */
  p->vc0_range = 0.1;                        /* :61 */
  p->qc0_range = radians(1.0);               /* :62 */
  p->wc0_range = radians(0.1);               /* :63 */
  p->qt0_range = radians(45.0);              /* :64 */
  p->wt0_range = radians(3.0);               /* :65 */
  p->dt = 1.0;                               /* :69 */
  p->t_max = 120.0;                          /* :70 */
  const double m = 100.0;                    /* :74 */
  const double inertia = 1.0 * 1.0 / 12.0 * m * (2.0 * 1.0); /* :75-79 */
  p->max_delta_v = 10.0 / m * 0.5;           /* :81 */
  p->max_delta_w = 0.2 / inertia * 0.5;      /* :82 */
  p->max_axial_distance = norm3(p->nominal_rc0) + 10.0; /* :85 */
  p->max_axial_speed = 5.0;                  /* :86 */
  p->max_wc = radians(10.0);                 /* :87 */
  p->max_attitude_error = radians(30.0);     /* :89 */
  p->koz_radius = 5.0;                       /* :93 */
  p->corridor_half_angle = radians(30.0);    /* :94 */
  p->corridor_axis[1] = -1.0;                /* :95 */
  p->capture_axis[1] = 1.0;                  /* :73 */
  p->rd[1] = -2.0;                           /* :104 */
  p->max_rd_error = 0.5;                     /* :105 */
  p->max_vd_error = 0.1;                     /* :106 */
  p->max_qd_error = radians(5.0);            /* :107 */
  p->max_wd_error = radians(1.0);            /* :108 */
  p->bubble_radius0 = p->max_axial_distance; /* :114 */
  p->bubble_decrease_rate = 0.5 * p->dt;     /* :115 */
  p->bubble_min = norm3(p->rd) + 2.0 * p->max_rd_error; /* :116 */
  const double mu = 3.986004418e14, Re = 6371e3, h = 800e3; /* :122-124 */
  const double ro = Re + h;                  /* :125 */
  p->n = sqrt(mu / (ro * ro * ro));          /* :126 */
  p->collision_coef = 0.5; p->bonus_coef = 8.0; p->fuel_coef = 0.2; p->att_coef = 1.0; /* :313 */
}

/* ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. 2011, public algorithm) */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
}

void orc_philox_block(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t block, uint32_t out[4]) {
  uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode, block};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  memcpy(out, c, sizeof c);
}

/* 24 uniforms from 4 blocks: every 64-bit half of a block yields three 21-bit fields, u = (field + 0.5) / 2^21. */
void orc_philox_uniforms(uint64_t seed, uint64_t env_id, uint32_t episode, double u[24]) {
  for (uint32_t j = 0; j < 4; ++j) {
    uint32_t c[4];
    orc_philox_block(seed, env_id, episode, j, c);
    for (int h = 0; h < 2; ++h) {
      const uint32_t lo = c[2 * h], hi = c[2 * h + 1];
      const uint32_t f0 = lo & 0x1FFFFFu, f1 = ((lo >> 21) | (hi << 11)) & 0x1FFFFFu, f2 = (hi >> 10) & 0x1FFFFFu;
      u[6 * j + 3 * h + 0] = ((double)f0 + 0.5) * (1.0 / 2097152.0);
      u[6 * j + 3 * h + 1] = ((double)f1 + 0.5) * (1.0 / 2097152.0);
      u[6 * j + 3 * h + 2] = ((double)f2 + 0.5) * (1.0 / 2097152.0);
    }
  }
}

/* ---------------------------------------------------------------- utils/quaternions.py */
void orc_quat2mat(const double qin[4], double m[9]) { /* quaternions.py:48-68 */
  const double mag = norm4(qin);                     /* :57 q = q / |q| */
  const double qw = qin[0] / mag, qx = qin[1] / mag, qy = qin[2] / mag, qz = qin[3] / mag;
  m[0] = 2 * (qw * qw + qx * qx) - 1; m[1] = 2 * (qx * qy - qw * qz);     m[2] = 2 * (qx * qz + qw * qy);
  m[3] = 2 * (qx * qy + qw * qz);     m[4] = 2 * (qw * qw + qy * qy) - 1; m[5] = 2 * (qy * qz - qw * qx);
  m[6] = 2 * (qx * qz - qw * qy);     m[7] = 2 * (qy * qz + qw * qx);     m[8] = 2 * (qw * qw + qz * qz) - 1;
}

void orc_rot2quat(const double axis_in[3], double theta, double q[4]) { /* quaternions.py:11-27 */
  const double an = norm3(axis_in);
  const double axis[3] = {axis_in[0] / an, axis_in[1] / an, axis_in[2] / an};   /* :23 */
  const double s = sin(theta / 2);
  q[0] = cos(theta / 2); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s; /* :24 */
  const double mag = norm4(q);                                                   /* :26 */
  for (int i = 0; i < 4; ++i) q[i] /= mag;
}

void orc_quat_product(const double q1in[4], const double q2in[4], double out[4]) { /* quaternions.py:149-170 */
  const double m1 = norm4(q1in), m2 = norm4(q2in);   /* :159-160 inputs normalised, output is not */
  const double q1[4] = {q1in[0] / m1, q1in[1] / m1, q1in[2] / m1, q1in[3] / m1};
  const double q2[4] = {q2in[0] / m2, q2in[1] / m2, q2in[2] / m2, q2in[3] / m2};
  const double s1 = q1[0], s2 = q2[0];
  const double* v1 = q1 + 1; const double* v2 = q2 + 1;
  double c[3]; cross3(v1, v2, c);
  out[0] = s1 * s2 - dot3(v1, v2);                   /* :165 */
  for (int i = 0; i < 3; ++i) out[1 + i] = s1 * v2[i] + s2 * v1[i] + c[i]; /* :166 */
}

/* ---------------------------------------------------------------- utils/general.py */
double orc_angle_between(const double a[3], const double b[3]) { /* general.py:163-181 */
  const double c = dot3(a, b) / (norm3(a) * norm3(b));
  /* :179 round(c, 5) on a np.float64 == np.round: multiply, rint (half to even), divide */
  const double r = rint(c * 1e5) / 1e5;
  return acos(r);
}

static void random_unit_vector(const double u3[3], double out[3]) { /* general.py:248-254 */
  const double v[3] = {-1.0 + 2.0 * u3[0], -1.0 + 2.0 * u3[1], -1.0 + 2.0 * u3[2]}; /* uniform(-1,1) = low + (high-low)*u */
  const double nv = norm3(v);
  for (int i = 0; i < 3; ++i) out[i] = v[i] / nv;
}

static double normalize_value(double val, double low, double high) { /* general.py:230-245, custom_range [-1, 1] */
  const double a = -1.0, b = 1.0;
  return (b - a) * (val - low) / (high - low) + a;
}

/* ---------------------------------------------------------------- utils/dynamics.py */
void orc_cw_solution(const double r0[3], const double v0[3], double n, double t, double r[3], double v[3]) {
  /* dynamics.py:24-55: closed-form Clohessy-Wiltshire state transition matrix, rows as at :40-47 */
  const double x0[6] = {r0[0], r0[1], r0[2], v0[0], v0[1], v0[2]};
  const double nt = n * t, c = cos(nt), s = sin(nt);
  const double stm[6][6] = {
      {4 - 3 * c,        0, 0,      1 / n * s,        2 / n * (1 - c),          0},
      {6 * (s - nt),     1, 0,      -2 / n * (1 - c), 1 / n * (4 * s - 3 * nt), 0},
      {0,                0, c,      0,                0,                        1 / n * s},
      {3 * n * s,        0, 0,      c,                2 * s,                    0},
      {-6 * n * (1 - c), 0, 0,      -2 * s,           4 * c - 3,                0},
      {0,                0, -n * s, 0,                0,                        c}};
  double xt[6];
  for (int i = 0; i < 6; ++i) {
    double acc = 0;
    for (int j = 0; j < 6; ++j) acc += stm[i][j] * x0[j];
    xt[i] = acc;
  }
  r[0] = xt[0]; r[1] = xt[1]; r[2] = xt[2]; v[0] = xt[3]; v[1] = xt[4]; v[2] = xt[5];
}

void orc_att_rhs(const double y[7], double dy[7]) {
  /* dynamics.py:93-119 with the env's constant inertia (rendezvous_env.py:75-80, :96-101) and zero torque (:181,:184) */
  const double inertia = 1.0 * 1.0 / 12.0 * 100.0 * 2.0;
  const double inv_inertia = 1.0 / inertia;
  const double mag = norm4(y);                        /* dynamics.py:109 */
  const double q0[4] = {y[0] / mag, y[1] / mag, y[2] / mag, y[3] / mag};
  const double mag2 = norm4(q0);                      /* dynamics.py:134 (quat_derivative normalises again) */
  const double q[4] = {q0[0] / mag2, q0[1] / mag2, q0[2] / mag2, q0[3] / mag2};
  const double w1 = y[4], w2 = y[5], w3 = y[6];
  /* dynamics.py:137-142 skew matrix for body-frame rates, :151 q_dot = 0.5 * skew @ q */
  dy[0] = 0.5 * (-w1 * q[1] - w2 * q[2] - w3 * q[3]);
  dy[1] = 0.5 * (w1 * q[0] + w3 * q[2] - w2 * q[3]);
  dy[2] = 0.5 * (w2 * q[0] - w3 * q[1] + w1 * q[3]);
  dy[3] = 0.5 * (w3 * q[0] + w2 * q[1] - w1 * q[2]);
  /* dynamics.py:169-171 Euler's equations */
  const double w[3] = {w1, w2, w3};
  const double L[3] = {inertia * w1, inertia * w2, inertia * w3};
  double cp[3]; cross3(w, L, cp);
  for (int i = 0; i < 3; ++i) dy[4 + i] = inv_inertia * (0.0 - cp[i]);
}

void orc_att_rhs_general(const double y[7], const double inertia[9], const double inv_inertia[9], const double torque[3],
                         double dy[7]) {
  /* derivative_of_att_and_rot_rate (dynamics.py:93-119) for any inertia tensor (row-major 3x3) and body torque */
  const double mag = norm4(y);                        /* dynamics.py:109 */
  const double q0[4] = {y[0] / mag, y[1] / mag, y[2] / mag, y[3] / mag};
  const double mag2 = norm4(q0);                      /* dynamics.py:134 (quat_derivative normalises again) */
  const double q[4] = {q0[0] / mag2, q0[1] / mag2, q0[2] / mag2, q0[3] / mag2};
  const double w1 = y[4], w2 = y[5], w3 = y[6];
  dy[0] = 0.5 * (-w1 * q[1] - w2 * q[2] - w3 * q[3]);   /* dynamics.py:137-151 */
  dy[1] = 0.5 * (w1 * q[0] + w3 * q[2] - w2 * q[3]);
  dy[2] = 0.5 * (w2 * q[0] - w3 * q[1] + w1 * q[3]);
  dy[3] = 0.5 * (w3 * q[0] + w2 * q[1] - w1 * q[2]);
  /* dynamics.py:169-171: w_dot = inv_inertia @ (torque - w x (inertia @ w)) */
  const double w[3] = {w1, w2, w3};
  double L[3], cp[3], rhs[3];
  for (int i = 0; i < 3; ++i) L[i] = inertia[3 * i] * w1 + inertia[3 * i + 1] * w2 + inertia[3 * i + 2] * w3;
  cross3(w, L, cp);
  for (int i = 0; i < 3; ++i) rhs[i] = torque[i] - cp[i];
  for (int i = 0; i < 3; ++i)
    dy[4 + i] = inv_inertia[3 * i] * rhs[0] + inv_inertia[3 * i + 1] * rhs[1] + inv_inertia[3 * i + 2] * rhs[2];
}

/* Dormand-Prince 5(4) with scipy's step-size control (scipy/integrate/_ivp/rk.py, common.py; public algorithm), as
 * solve_ivp(method='RK45', rtol, atol) runs it for rendezvous_env.py:561-570 / :588-597 (rtol=1e-7, atol=1e-6 there).
 * body == NULL: the env's constant isotropic tensor and zero torque (orc_att_rhs); otherwise the general right-hand side. */
typedef struct RigidCtx { const double* inertia; const double* inv_inertia; const double* torque; } RigidCtx;
static void att_rhs(const RigidCtx* body, const double y[7], double dy[7]) {
  if (body) orc_att_rhs_general(y, body->inertia, body->inv_inertia, body->torque, dy);
  else orc_att_rhs(y, dy);
}
static double rms7(const double* x) {
  double s = 0; for (int i = 0; i < 7; ++i) s += x[i] * x[i];
  return sqrt(s) / sqrt(7.0);
}
static int rk45_integrate(double y[7], double t_bound, const RigidCtx* body, double rtol, double atol) {
  static const double A[6][5] = {{0, 0, 0, 0, 0},
                                 {1.0 / 5, 0, 0, 0, 0},
                                 {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                 {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                 {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                 {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
  static const double B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
  static const double E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
  double t = 0, f[7], K[7][7], scale[7], tmp[7];
  int nfev = 0, attempts = 0;   /* attempts: accepted + rejected steps; capped (scipy has no cap; 4096 per dt means it is failing) */
  att_rhs(body, y, f); ++nfev;
  /* select_initial_step, order = 4 */
  double h_abs;
  {
    double a[7], b[7];
    for (int i = 0; i < 7; ++i) { scale[i] = atol + fabs(y[i]) * rtol; a[i] = y[i] / scale[i]; b[i] = f[i] / scale[i]; }
    const double d0 = rms7(a), d1 = rms7(b);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    if (h0 > t_bound) h0 = t_bound;
    double y1[7], f1[7];
    for (int i = 0; i < 7; ++i) y1[i] = y[i] + h0 * f[i];
    att_rhs(body, y1, f1); ++nfev;
    for (int i = 0; i < 7; ++i) a[i] = (f1[i] - f[i]) / scale[i];
    const double d2 = rms7(a) / h0;
    double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
    h_abs = fmin(fmin(100 * h0, h1), t_bound);
  }
  while (t < t_bound) {
    const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
    if (h_abs < min_step) h_abs = min_step;
    int rejected = 0;
    for (;;) {
      double h = h_abs, t_new = t + h;
      if (t_new - t_bound > 0) t_new = t_bound;
      h = t_new - t;
      h_abs = fabs(h);
      for (int i = 0; i < 7; ++i) K[0][i] = f[i];
      for (int s = 1; s < 6; ++s) {
        for (int i = 0; i < 7; ++i) {
          double dy = 0;
          for (int j = 0; j < s; ++j) dy += K[j][i] * A[s][j];
          tmp[i] = y[i] + dy * h;
        }
        att_rhs(body, tmp, K[s]); ++nfev;
      }
      double y_new[7], f_new[7];
      for (int i = 0; i < 7; ++i) {
        double acc = 0;
        for (int s = 0; s < 6; ++s) acc += K[s][i] * B[s];
        y_new[i] = y[i] + h * acc;
      }
      att_rhs(body, y_new, f_new); ++nfev;
      for (int i = 0; i < 7; ++i) K[6][i] = f_new[i];
      double err[7];
      for (int i = 0; i < 7; ++i) {
        double acc = 0;
        for (int s = 0; s < 7; ++s) acc += K[s][i] * E[s];
        const double sc = atol + fmax(fabs(y[i]), fabs(y_new[i])) * rtol;
        err[i] = acc * h / sc;
      }
      const double error_norm = rms7(err);
      ++attempts;
      if (error_norm < 1) {
        double factor = error_norm == 0 ? 10.0 : fmin(10.0, 0.9 * pow(error_norm, -0.2));
        if (rejected && factor > 1) factor = 1;
        h_abs *= factor;
        t = t_new;
        for (int i = 0; i < 7; ++i) { y[i] = y_new[i]; f[i] = f_new[i]; }
        if (attempts >= 4096 && t < t_bound) { for (int i = 0; i < 7; ++i) y[i] = NAN; return nfev; }
        break;
      }
      if (!(error_norm == error_norm) || h_abs <= min_step || attempts >= 4096) {
        /* scipy gives up here ("step size too small" / non-finite state) and the reference then fails on the empty
         * solution (rendezvous_env.py:572-574); the restatement poisons the state instead of looping */
        for (int i = 0; i < 7; ++i) y[i] = NAN;
        return nfev;
      }
      h_abs *= fmax(0.2, 0.9 * pow(error_norm, -0.2));
      if (h_abs < min_step) h_abs = min_step;
      rejected = 1;
    }
  }
  return nfev;
}

int orc_solve_attitude_rk45(double y[7], double dt, const double inertia[9], const double inv_inertia[9], const double torque[3],
                            double rtol, double atol) {
  const RigidCtx body = {inertia, inv_inertia, torque};
  return rk45_integrate(y, dt, &body, rtol, atol);
}

void orc_integrate_attitude_general(double q[4], double w[3], double dt, const double inertia[9], const double inv_inertia[9],
                                    const double torque[3], double rtol, double atol) {
  /* rendezvous_env.py:552-604 with self.inertia / self.inv_inertia (or the target's) and the torque argument as given */
  double y[7] = {q[0], q[1], q[2], q[3], w[0], w[1], w[2]};
  (void)orc_solve_attitude_rk45(y, dt, inertia, inv_inertia, torque, rtol, atol);
  const double mag = norm4(y);                                 /* :574, :601 */
  for (int i = 0; i < 4; ++i) q[i] = y[i] / mag;
  for (int i = 0; i < 3; ++i) w[i] = y[4 + i];
}

void orc_integrate_attitude(double q[4], double w[3], double dt, int integrator) {
  /* rendezvous_env.py:552-604: y0=[q,w] -> solve_ivp(RK45) over [0,dt] -> q /= |q| (:574, :601). */
  if (integrator == ORC_INTEGRATOR_RK45) {
    double y[7] = {q[0], q[1], q[2], q[3], w[0], w[1], w[2]};
    (void)rk45_integrate(y, dt, NULL, 1e-7, 1e-6);   /* :567-568 */
    const double mag = norm4(y);
    for (int i = 0; i < 4; ++i) q[i] = y[i] / mag;
    for (int i = 0; i < 3; ++i) w[i] = y[4 + i];
    return;
  }
  /* Exact solution: inertia = c*Identity and torque = 0 make w x (I w) = 0 (dynamics.py:169-171), so w is constant and
   * q_dot = 0.5 * Omega(w) q (dynamics.py:137-151) integrates to the right-multiplication q (x) [cos(|w|dt/2), w_hat sin(|w|dt/2)]. */
  /* The right-hand side normalises q before differentiating it (dynamics.py:109, :134) but the integrated state is the quaternion
   * as given: for |q| = rho the solution keeps its norm and turns at w / rho.  Every step ends with a normalisation (:574, :601),
   * so rho differs from 1 only for a state injected from outside (monte_carlo.py:107-112 normalises first, :66-67). */
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (fabs(n2 - 1.0) > 1e-6) dt = dt / sqrt(n2);
  const double wn = norm3(w);
  const double half = 0.5 * wn * dt;
  double dq[4];
  dq[0] = cos(half);
  /* sin(half)/wn with the limit 0.5*dt at wn -> 0 */
  const double k = wn > 0 ? sin(half) / wn : 0.5 * dt;
  dq[1] = w[0] * k; dq[2] = w[1] * k; dq[3] = w[2] * k;
  const double a = q[0], b = q[1], c = q[2], d = q[3];
  double o[4];
  o[0] = a * dq[0] - b * dq[1] - c * dq[2] - d * dq[3];
  o[1] = a * dq[1] + b * dq[0] + c * dq[3] - d * dq[2];
  o[2] = a * dq[2] - b * dq[3] + c * dq[0] + d * dq[1];
  o[3] = a * dq[3] + b * dq[2] - c * dq[1] + d * dq[0];
  const double mag = norm4(o);
  for (int i = 0; i < 4; ++i) q[i] = o[i] / mag;
}

/* ---------------------------------------------------------------- rendezvous_env.py helpers */
static void chaser2lvlh(const OrcEnv* e, const double v[3], double out[3]) { double m[9]; orc_quat2mat(e->qc, m); matvec(m, v, out); }  /* :490-498 */
static void target2lvlh(const OrcEnv* e, const double v[3], double out[3]) { double m[9]; orc_quat2mat(e->qt, m); matvec(m, v, out); }  /* :500-508 */
static void lvlh2chaser(const OrcEnv* e, const double v[3], double out[3]) { double m[9]; orc_quat2mat(e->qc, m); matTvec(m, v, out); } /* :470-478 */
static void lvlh2target(const OrcEnv* e, const double v[3], double out[3]) { double m[9]; orc_quat2mat(e->qt, m); matTvec(m, v, out); } /* :480-488 */

static double get_attitude_error(const OrcParams* p, const OrcEnv* e) { /* :424-434 */
  double cap[3]; chaser2lvlh(e, p->capture_axis, cap);
  const double neg_rc[3] = {-e->rc[0], -e->rc[1], -e->rc[2]};
  return orc_angle_between(neg_rc, cap);
}

void orc_get_observation(const OrcParams* p, const OrcEnv* e, float obs[17]) { /* :294-311 */
  for (int i = 0; i < 3; ++i) obs[i] = (float)normalize_value(e->rc[i], -p->max_axial_distance, p->max_axial_distance);
  for (int i = 0; i < 3; ++i) obs[3 + i] = (float)normalize_value(e->vc[i], -p->max_axial_speed, p->max_axial_speed);
  for (int i = 0; i < 4; ++i) obs[6 + i] = (float)e->qc[i];
  for (int i = 0; i < 3; ++i) obs[10 + i] = (float)normalize_value(e->wc[i], -p->max_wc, p->max_wc);
  for (int i = 0; i < 4; ++i) obs[13 + i] = (float)e->qt[i];   /* wt is not observed (:309) */
}

void orc_get_errors(const OrcParams* p, const OrcEnv* e, double err[4]) { /* :451-468 */
  double wc_l[3], wt_l[3], rd_l[3], vd_l[3];
  chaser2lvlh(e, e->wc, wc_l);           /* :458 */
  target2lvlh(e, e->wt, wt_l);           /* :459 */
  target2lvlh(e, p->rd, rd_l);           /* :460 */
  cross3(wt_l, rd_l, vd_l);              /* :461 */
  const double dp[3] = {e->rc[0] - rd_l[0], e->rc[1] - rd_l[1], e->rc[2] - rd_l[2]};
  const double dv[3] = {e->vc[0] - vd_l[0], e->vc[1] - vd_l[1], e->vc[2] - vd_l[2]};
  const double dw[3] = {wc_l[0] - wt_l[0], wc_l[1] - wt_l[1], wc_l[2] - wt_l[2]};
  err[0] = norm3(dp);                    /* :463 */
  err[1] = norm3(dv);                    /* :464 */
  err[2] = get_attitude_error(p, e);     /* :465 */
  err[3] = norm3(dw);                    /* :466 */
}

int orc_check_collision(const OrcParams* p, const OrcEnv* e) { /* :388-404 */
  if (norm3(e->rc) < p->koz_radius) {
    double ax[3]; target2lvlh(e, p->corridor_axis, ax);
    if (orc_angle_between(e->rc, ax) > p->corridor_half_angle) return 1;
  }
  return 0;
}

int orc_check_success(const OrcParams* p, const OrcEnv* e) { /* :406-422 */
  if (e->collided) return 0;
  double err[4]; orc_get_errors(p, e, err);
  return (err[0] <= p->max_rd_error && err[1] <= p->max_vd_error && err[2] <= p->max_qd_error && err[3] <= p->max_wd_error) ? 1 : 0;
}

double orc_dist_from_koz(const OrcParams* p, const OrcEnv* e) { /* :510-537 */
  const double pos_mag = norm3(e->rc), r_koz = p->koz_radius, th_c = p->corridor_half_angle;
  double ax[3]; target2lvlh(e, p->corridor_axis, ax);
  const double th = orc_angle_between(e->rc, ax);
  if (pos_mag < r_koz) {
    if (th >= th_c) {
      const double d_rad = r_koz - pos_mag;
      const double d_tan = pos_mag * sin(fmin(th - th_c, ORC_PI / 2));
      return -1 * fmin(d_rad, d_tan);
    }
    return pos_mag * sin(th_c - th);
  }
  if (th >= th_c) return pos_mag - r_koz;
  const double d_rad = pos_mag - r_koz * cos(th_c - th);
  const double d_tan = r_koz * sin(th_c - th);
  return sqrt(d_rad * d_rad + d_tan * d_tan);
}

void orc_diagnose(const OrcParams* p, const OrcEnv* e, double diag[8]) {
  orc_get_errors(p, e, diag);
  diag[4] = orc_check_collision(p, e);
  diag[5] = orc_check_success(p, e);
  diag[6] = orc_dist_from_koz(p, e);
  diag[7] = e->collided;
}

static double env_time(const OrcParams* p, const OrcEnv* e) {
  /* :193 t = round(t + dt, 3) every step == the decimal k*dt rounded to 3 places */
  return rint((double)e->k * p->dt * 1e3) / 1e3;
}

static void canon_state(OrcEnv* e, int storage) {
  if (storage != ORC_STORAGE_F32) return;
  double* s = e->rc; /* rc..wt are contiguous 20 doubles */
  for (int i = 0; i < 20; ++i) s[i] = (double)(float)s[i];
}

/* ---------------------------------------------------------------- reset: rendezvous_env.py:223-270 */
static void reset_one(const OrcParams* p, const OrcConfig* c, OrcEnv* e, int64_t i, int64_t n) {
  if (c->tape_depth > 0 && c->tape) {
    const double* s = c->tape + (((int64_t)(e->episode % c->tape_depth) * n) + i) * 20;
    memcpy(e->rc, s, 20 * sizeof(double));
  } else {
    double u[24];
    orc_philox_uniforms(c->seed, c->env_id_offset + (uint64_t)i, (uint32_t)e->episode, u);
    double dir[3], rc_dev[3], vc_dev[3], wc_dev[3], wt_dev[3], qc_dev[4], qt_dev[4];
    random_unit_vector(u + 0, dir);  for (int j = 0; j < 3; ++j) rc_dev[j] = dir[j] * (0.0 + (p->rc0_range - 0.0) * u[3]);   /* :231 */
    random_unit_vector(u + 4, dir);  for (int j = 0; j < 3; ++j) vc_dev[j] = dir[j] * (0.0 + (p->vc0_range - 0.0) * u[7]);   /* :234 */
    const double theta_c = 0.0 + (p->qc0_range - 0.0) * u[8];                                                               /* :237 */
    random_unit_vector(u + 9, dir);  orc_rot2quat(dir, theta_c, qc_dev);                                                      /* :238-239 */
    random_unit_vector(u + 12, dir); for (int j = 0; j < 3; ++j) wc_dev[j] = dir[j] * (0.0 + (p->wc0_range - 0.0) * u[15]); /* :242 */
    const double theta_t = 0.0 + (p->qt0_range - 0.0) * u[16];                                                              /* :245 */
    random_unit_vector(u + 17, dir); orc_rot2quat(dir, theta_t, qt_dev);                                                     /* :246-247 */
    random_unit_vector(u + 20, dir); for (int j = 0; j < 3; ++j) wt_dev[j] = dir[j] * (0.0 + (p->wt0_range - 0.0) * u[23]); /* :250 */
    for (int j = 0; j < 3; ++j) e->rc[j] = p->nominal_rc0[j] + rc_dev[j];      /* :253 */
    for (int j = 0; j < 3; ++j) e->vc[j] = p->nominal_vc0[j] + vc_dev[j];      /* :254 */
    orc_quat_product(qc_dev, p->nominal_qc0, e->qc);                           /* :255 */
    double tmp[3];
    for (int j = 0; j < 3; ++j) tmp[j] = p->nominal_wc0[j] + wc_dev[j];
    lvlh2chaser(e, tmp, e->wc);                                                /* :256 */
    orc_quat_product(qt_dev, p->nominal_qt0, e->qt);                           /* :257 */
    for (int j = 0; j < 3; ++j) tmp[j] = p->nominal_wt0[j] + wt_dev[j];
    lvlh2target(e, tmp, e->wt);                                                /* :258 */
  }
  canon_state(e, c->storage);
  e->collided = orc_check_collision(p, e);        /* :261 */
  e->success = orc_check_success(p, e);           /* :262 */
  e->bubble_radius = canon(p->bubble_radius0, c->storage); /* :263 */
  e->total_delta_v = 0; e->total_delta_w = 0;     /* :264-265 */
  e->k = 0;                                       /* :266 */
  e->episode_return = 0;
  e->halted = 0;
  e->episode += 1;
}

void orc_reset(const OrcParams* p, const OrcConfig* c, int64_t n, OrcEnv* envs, const uint8_t* mask, float* obs_out) {
  for (int64_t i = 0; i < n; ++i) {
    if (mask && !mask[i]) continue;
    reset_one(p, c, &envs[i], i, n);
    if (obs_out) orc_get_observation(p, &envs[i], obs_out + 17 * i);
  }
}

/* ---------------------------------------------------------------- step: rendezvous_env.py:160-221 */
typedef struct StepLocal { int done, reason, success_ep, collided_ep; double ret, len, dv, dw; } StepLocal;

static void step_one(const OrcParams* p, const OrcConfig* c, OrcEnv* e, int64_t i, int64_t n, const float* a32,
                     const OrcStepOut* out, StepLocal* loc) {
  loc->done = 0; loc->reason = 0;
  if (e->halted) {
    orc_get_observation(p, e, out->obs + 17 * i);
    out->reward[i] = 0; out->done[i] = 1;
    if (out->done_reason) out->done_reason[i] = 0;
    if (out->diag) orc_diagnose(p, e, out->diag + 8 * i);
    loc->done = -1;
    return;
  }
  /* :172 float32 array * Python float stays float32 */
  const float dvb32[3] = {a32[0] * (float)p->max_delta_v, a32[1] * (float)p->max_delta_v, a32[2] * (float)p->max_delta_v};
  const float dwb32[3] = {a32[3] * (float)p->max_delta_w, a32[4] * (float)p->max_delta_w, a32[5] * (float)p->max_delta_w};
  /* (dwb32 is the NumPy-1.x float32 product; NEP 50 promotes a[3:] * np.float64 to float64, see header) */
  const double dvb[3] = {dvb32[0], dvb32[1], dvb32[2]};
  double delta_v[3]; chaser2lvlh(e, dvb, delta_v);                                          /* :172 */
  const double v_imp[3] = {e->vc[0] + delta_v[0], e->vc[1] + delta_v[1], e->vc[2] + delta_v[2]}; /* :176 */
  double rn[3], vn[3];
  orc_cw_solution(e->rc, v_imp, p->n, p->dt, rn, vn);                                       /* :177 */
  memcpy(e->rc, rn, sizeof rn); memcpy(e->vc, vn, sizeof vn);
  for (int j = 0; j < 3; ++j)                                                               /* :173, :180 */
    e->wc[j] = e->wc[j] + (c->numpy_legacy ? (double)dwb32[j] : (double)a32[3 + j] * p->max_delta_w);
  if (c->integrator == ORC_INTEGRATOR_GENERAL && c->rigid) {
    const OrcRigidBody* b = c->rigid;
    orc_integrate_attitude_general(e->qc, e->wc, p->dt, b->inertia_chaser, b->inv_inertia_chaser, b->torque_chaser, b->rtol, b->atol); /* :181 */
    orc_integrate_attitude_general(e->qt, e->wt, p->dt, b->inertia_target, b->inv_inertia_target, b->torque_target, b->rtol, b->atol); /* :184 */
  } else {
    orc_integrate_attitude(e->qc, e->wc, p->dt, c->integrator);                             /* :181 */
    orc_integrate_attitude(e->qt, e->wt, p->dt, c->integrator);                             /* :184 */
  }
  canon_state(e, c->storage);

  if (!e->collided) {                                                                       /* :187-190 */
    e->collided = orc_check_collision(p, e);
    if (orc_check_success(p, e)) e->success += 1;
  }
  e->k += 1;                                                                                /* :193 */
  const double t = env_time(p, e);
  e->bubble_radius -= p->bubble_decrease_rate;                                              /* :196-198 */
  if (e->bubble_radius < p->bubble_min) e->bubble_radius = p->bubble_min;
  e->bubble_radius = canon(e->bubble_radius, c->storage);
  /* :201-202 float32 sums (np.abs(..).sum() of a float32 array); promotion of the products: see header */
  const float sum_v32 = (fabsf(a32[0]) + fabsf(a32[1])) + fabsf(a32[2]);
  const float sum_w32 = (fabsf(a32[3]) + fabsf(a32[4])) + fabsf(a32[5]);
  if (c->numpy_legacy) {
    e->total_delta_v = canon(e->total_delta_v + (double)sum_v32 * p->max_delta_v, c->storage);
  } else {
    const float inc = sum_v32 * (float)p->max_delta_v;
    e->total_delta_v = (double)((float)e->total_delta_v + inc);   /* 0 + np.float32 stays np.float32 */
  }
  e->total_delta_w = canon(e->total_delta_w + (double)sum_w32 * p->max_delta_w, c->storage);

  float obs[17]; orc_get_observation(p, e, obs);                                            /* :205 */

  /* :355-386 get_done_condition */
  const double dist = norm3(e->rc);
  const double att = get_attitude_error(p, e);
  int outside = 0;
  for (int j = 0; j < 17; ++j) if (!(obs[j] >= -1.0f && obs[j] <= 1.0f)) outside = 1;     /* Box.contains, NaN -> outside */
  const int conds[4] = {outside, t >= p->t_max, dist > e->bubble_radius, att > p->max_attitude_error};
  int done = 0, reason = 0;
  for (int j = 3; j >= 0; --j) if (conds[j]) { done = 1; reason = j + 1; }                  /* :381 first true */

  /* :313-353 get_bubble_reward */
  double rew = 0;
  rew += p->dt * p->att_coef * (1 - att / p->max_attitude_error);                          /* :329 */
  if (c->numpy_legacy)                                                                     /* :333 (the sign is + in the reference) */
    rew += p->dt * p->fuel_coef * (double)sum_v32 / (3 * p->max_delta_v);
  else
    rew += (double)(((float)(p->dt * p->fuel_coef) * sum_v32) / (float)(3 * p->max_delta_v));
  if (orc_check_collision(p, e)) rew -= p->dt * p->collision_coef;                         /* :336-337 */
  if (dist < p->koz_radius && !e->collided) {                                              /* :340 */
    double err[4]; orc_get_errors(p, e, err);
    if (err[0] < p->max_rd_error) {                                                        /* :348-351 */
      rew += p->dt * p->bonus_coef * (2 - err[0] / p->max_rd_error);
      if (err[2] < p->max_qd_error) rew += p->dt * p->bonus_coef * (2 - err[2] / p->max_qd_error);
    }
  }
  e->episode_return = canon(e->episode_return + rew, c->storage);

  out->reward[i] = rew; out->done[i] = (uint8_t)done;
  if (out->done_reason)   /* reason | collided-in-episode << 4 | succeeded-in-episode << 5 (flag bits only where done) */
    out->done_reason[i] = (uint8_t)(reason | ((done && e->collided) ? 16 : 0) | ((done && e->success > 0) ? 32 : 0));
  if (out->diag) orc_diagnose(p, e, out->diag + 8 * i);
  loc->done = done; loc->reason = reason;
  if (done) {
    if (out->terminal_obs) memcpy(out->terminal_obs + 17 * i, obs, sizeof obs);
    if (out->episode_return) out->episode_return[i] = e->episode_return;
    if (out->episode_length) out->episode_length[i] = e->k;
    loc->ret = e->episode_return; loc->len = e->k; loc->dv = e->total_delta_v; loc->dw = e->total_delta_w;
    loc->success_ep = e->success > 0; loc->collided_ep = e->collided != 0;
    if (c->on_done == ORC_ON_DONE_RESET) {
      reset_one(p, c, e, i, n);
      orc_get_observation(p, e, obs);   /* SB3 DummyVecEnv: the returned obs is the reset obs */
    } else if (c->on_done == ORC_ON_DONE_HALT) {
      e->halted = 1;
    }
  }
  memcpy(out->obs + 17 * i, obs, sizeof obs);
}

void orc_step(const OrcParams* p, const OrcConfig* c, int64_t n, OrcEnv* envs, const float* actions,
              const OrcStepOut* out, OrcStats* stats, int n_threads) {
  OrcStats acc; memset(&acc, 0, sizeof acc);
  uint64_t steps = 0, episodes = 0, succ = 0, coll = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  double sret = 0, slen = 0, sdv = 0, sdw = 0;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static) \
    reduction(+ : steps, episodes, succ, coll, r0, r1, r2, r3, sret, slen, sdv, sdw)
#endif
  for (int64_t i = 0; i < n; ++i) {
    StepLocal loc;
    step_one(p, c, &envs[i], i, n, actions + 6 * i, out, &loc);
    if (loc.done >= 0) steps += 1;
    if (loc.done > 0) {
      episodes += 1; succ += loc.success_ep; coll += loc.collided_ep;
      r0 += loc.reason == 1; r1 += loc.reason == 2; r2 += loc.reason == 3; r3 += loc.reason == 4;
      sret += loc.ret; slen += loc.len; sdv += loc.dv; sdw += loc.dw;
    }
  }
  if (stats) {
    stats->env_steps += steps; stats->episodes += episodes; stats->successes += succ; stats->collisions += coll;
    stats->reasons[0] += r0; stats->reasons[1] += r1; stats->reasons[2] += r2; stats->reasons[3] += r3;
    stats->sum_return += sret; stats->sum_length += slen; stats->sum_delta_v += sdv; stats->sum_delta_w += sdw;
  }
}

/* ---------------------------------------------------------------- state access (monte_carlo.py:107-112) */
void orc_set_state(int64_t n, OrcEnv* envs, const double* states, int storage) {
  for (int64_t i = 0; i < n; ++i) {
    memcpy(envs[i].rc, states + 20 * i, 20 * sizeof(double));
    canon_state(&envs[i], storage);
  }
}
void orc_get_state(int64_t n, const OrcEnv* envs, double* states) {
  for (int64_t i = 0; i < n; ++i) memcpy(states + 20 * i, envs[i].rc, 20 * sizeof(double));
}
void orc_get_aux(const OrcParams* p, int64_t n, const OrcEnv* envs, double* aux) {
  for (int64_t i = 0; i < n; ++i) {
    const OrcEnv* e = &envs[i]; double* a = aux + 8 * i;
    a[0] = env_time(p, e); a[1] = e->bubble_radius; a[2] = e->collided; a[3] = e->success;
    a[4] = e->total_delta_v; a[5] = e->total_delta_w; a[6] = e->episode_return; a[7] = e->episode;
  }
}
void orc_observe(const OrcParams* p, int64_t n, const OrcEnv* envs, float* obs) {
  for (int64_t i = 0; i < n; ++i) orc_get_observation(p, &envs[i], obs + 17 * i);
}
void orc_diagnose_batch(const OrcParams* p, int64_t n, const OrcEnv* envs, double* diag) {
  for (int64_t i = 0; i < n; ++i) orc_diagnose(p, &envs[i], diag + 8 * i);
}
