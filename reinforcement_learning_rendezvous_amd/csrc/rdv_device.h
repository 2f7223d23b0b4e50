// rdv_device.h — per-lane device math of the fused rendezvous step for gfx950 (CDNA4, wave64).
//
// One lane owns one environment.  All arithmetic is fp64 (MI355X vector fp64 is half the fp32 rate and the path
// needs < 2 kFLOP per env-step, so precision is free relative to the launch; it removes the fp32 hazards of
// the 1e-5 rounding of the cosines in general.py:179 and the cancellation in the CW matrix).
// The persistent state is held in HBM in the storage type ST (float in production, double in parity mode) and is
// canonicalised to ST right after propagation, so that everything derived in the same step (flags, observation,
// reward, diagnostics) is a function of exactly the stored state.
//
// Citations are file:line in cfdeinza/reinforcement-learning-rendezvous.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdv {

// ---------------------------------------------------------------------------------------------------------------
// Kernel-argument parameter block (lives in SGPRs: it is wave-uniform).  Derived once on the host.
struct DevParams {
  // Clohessy-Wiltshire state-transition matrix for (n, dt), the 14 non-zeros of dynamics.py:40-47
  double phi_xx, phi_xvx, phi_xvy;      // row 0: 4-3c, s/n, 2(1-c)/n
  double phi_yx, phi_yvx, phi_yvy;      // row 1: 6(s-nt), -2(1-c)/n, (4s-3nt)/n     (phi_yy = 1)
  double phi_zz, phi_zvz;               // row 2: c, s/n
  double phi_vxx, phi_vxvx, phi_vxvy;   // row 3: 3ns, c, 2s
  double phi_vyx, phi_vyvx, phi_vyvy;   // row 4: -6n(1-c), -2s, 4c-3
  double phi_vzz, phi_vzvz;             // row 5: -ns, c
  double dt, half_dt, t_max;
  double max_delta_w;                   // np.float64 in the reference (:82) -> fp64 product (NEP 50)
  float  max_delta_v_f32;               // Python float * float32 array stays float32 (:172, :201)
  float  fuel_scale_f32;                // float32(dt * fuel_coef)  (:333)
  float  fuel_div_f32;                  // float32(3 * max_delta_v) (:333)
  float  pad0;
  double obs_lo_r, obs_span_r;          // -max_axial_distance, 2*max_axial_distance (normalize_value, general.py:243)
  double obs_lo_v, obs_span_v;
  double obs_lo_w, obs_span_w;
  double max_attitude_error, koz_radius, corridor_half_angle;
  double corridor_axis[3], capture_axis[3], rd[3];
  double max_rd_error, max_vd_error, max_qd_error, max_wd_error;
  double bubble_radius0, bubble_decrease_rate, bubble_min;
  double att_term, coll_term, bonus_term;   // dt*att_coef, dt*collision_coef, dt*bonus_coef
  // reset (rendezvous_env.py:229-258)
  double nominal_rc0[3], nominal_vc0[3], nominal_qc0[4], nominal_wc0[3], nominal_qt0[4], nominal_wt0[3];
  double rc0_range, vc0_range, qc0_range, wc0_range, qt0_range, wt0_range;
};

enum : uint32_t { FLAG_COLLIDED = 1u, FLAG_HALTED = 2u, SUCCESS_SHIFT = 2 };   // flags word: bit0, bit1, count << 2

// One environment in registers.
struct Env {
  double rc[3], vc[3], qc[4], wc[3], qt[4], wt[3];
  double bubble, sum_dv, sum_dw, ep_ret;
  int32_t k;
  uint32_t flags;
  uint32_t episode;
};

// Everything the flags / reward / done / diagnostics need from the (canonical) state.
struct Derived {
  double dist;       // |rc|
  double att;        // get_attitude_error (:424-434)
  double pos, vel, rot;   // get_errors (:451-468)
  double corr_cos;   // cos of the angle between rc and the corridor axis in LVLH, before rounding
};

__device__ __forceinline__ double canon(double x, float) { return (double)(float)x; }
__device__ __forceinline__ double canon(double x, double) { return x; }

// float32 multiply that is never contracted into an fma (NumPy rounds the product, then the sum)
__device__ __forceinline__ float mul_f32_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double norm3(const double* a) { return sqrt(dot3(a, a)); }

// quaternions.py:48-68 (normalises q first, :57)
__device__ __forceinline__ void quat2mat(const double* q, double* m) {
  const double mag = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double qw = q[0] / mag, qx = q[1] / mag, qy = q[2] / mag, qz = q[3] / mag;
  m[0] = 2 * (qw * qw + qx * qx) - 1; m[1] = 2 * (qx * qy - qw * qz);     m[2] = 2 * (qx * qz + qw * qy);
  m[3] = 2 * (qx * qy + qw * qz);     m[4] = 2 * (qw * qw + qy * qy) - 1; m[5] = 2 * (qy * qz - qw * qx);
  m[6] = 2 * (qx * qz - qw * qy);     m[7] = 2 * (qy * qz + qw * qx);     m[8] = 2 * (qw * qw + qz * qz) - 1;
}
__device__ __forceinline__ void matvec(const double* m, const double* v, double* o) {
  o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  o[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  o[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}
__device__ __forceinline__ void matTvec(const double* m, const double* v, double* o) {
  o[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  o[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  o[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}

// general.py:163-181: acos(round(cos, 5)); np.round == rint(x*1e5)/1e5 (ties to even)
__device__ __forceinline__ double rounded_acos(double c) { return acos(rint(c * 1e5) / 1e5); }

// rendezvous_env.py:552-604 with the reference's isotropic inertia and zero torque: w is constant and
// q(t+dt) = normalize(q (x) [cos(|w|dt/2), w_hat sin(|w|dt/2)])  (body-frame rate: right multiplication, dynamics.py:137-151)
__device__ __forceinline__ void integrate_attitude(double* q, const double* w, double half_dt, double dt) {
  const double wn = norm3(w);
  double s, c;
  sincos(wn * half_dt, &s, &c);
  const double k = wn > 0.0 ? s / wn : 0.5 * dt;
  const double dx = w[0] * k, dy = w[1] * k, dz = w[2] * k;
  const double a = q[0], b = q[1], cc = q[2], d = q[3];
  const double o0 = a * c - b * dx - cc * dy - d * dz;
  const double o1 = a * dx + b * c + cc * dz - d * dy;
  const double o2 = a * dy - b * dz + cc * c + d * dx;
  const double o3 = a * dz + b * dy - cc * dx + d * c;
  const double mag = sqrt(o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3);   // :574, :601
  q[0] = o0 / mag; q[1] = o1 / mag; q[2] = o2 / mag; q[3] = o3 / mag;
}

// Flags/reward inputs from the canonical state: one R(qc), one R(qt) per call (the reference rebuilds them ~10x per step).
__device__ __forceinline__ void derive(const DevParams& P, const Env& e, Derived& d, double* corr_l /*R_t * corridor_axis*/) {
  double Rc[9], Rt[9];
  quat2mat(e.qc, Rc);
  quat2mat(e.qt, Rt);
  double cap_l[3], wc_l[3], wt_l[3], rd_l[3];
  matvec(Rc, P.capture_axis, cap_l);    // :431
  matvec(Rc, e.wc, wc_l);               // :458
  matvec(Rt, e.wt, wt_l);               // :459
  matvec(Rt, P.rd, rd_l);               // :460
  matvec(Rt, P.corridor_axis, corr_l);  // :400
  const double vd_l[3] = {wt_l[1] * rd_l[2] - wt_l[2] * rd_l[1], wt_l[2] * rd_l[0] - wt_l[0] * rd_l[2],
                          wt_l[0] * rd_l[1] - wt_l[1] * rd_l[0]};                       // :461
  const double dp[3] = {e.rc[0] - rd_l[0], e.rc[1] - rd_l[1], e.rc[2] - rd_l[2]};
  const double dv[3] = {e.vc[0] - vd_l[0], e.vc[1] - vd_l[1], e.vc[2] - vd_l[2]};
  const double dw[3] = {wc_l[0] - wt_l[0], wc_l[1] - wt_l[1], wc_l[2] - wt_l[2]};
  d.dist = norm3(e.rc);
  d.pos = norm3(dp);                    // :463
  d.vel = norm3(dv);                    // :464
  d.rot = norm3(dw);                    // :466
  const double nrc[3] = {-e.rc[0], -e.rc[1], -e.rc[2]};
  d.att = rounded_acos(dot3(nrc, cap_l) / (d.dist * norm3(cap_l)));                     // :432
  d.corr_cos = dot3(e.rc, corr_l) / (d.dist * norm3(corr_l));                           // general.py:179 before rounding
}

// check_collision (:388-404): the rounded-cosine acos is only evaluated inside the KOZ sphere, as in the reference
__device__ __forceinline__ bool in_koz(const DevParams& P, const Derived& d) {
  bool c = false;
  if (d.dist < P.koz_radius) c = rounded_acos(d.corr_cos) > P.corridor_half_angle;
  return c;
}

__device__ __forceinline__ bool errors_ok(const DevParams& P, const Derived& d) {   // :416-417 (<=)
  return d.pos <= P.max_rd_error && d.vel <= P.max_vd_error && d.att <= P.max_qd_error && d.rot <= P.max_wd_error;
}

// dist_from_koz (:510-537)
__device__ __forceinline__ double dist_from_koz(const DevParams& P, const Derived& d) {
  const double pos_mag = d.dist, r_koz = P.koz_radius, th_c = P.corridor_half_angle;
  const double th = rounded_acos(d.corr_cos);
  const double pi_2 = 1.57079632679489661923;
  double out;
  if (pos_mag < r_koz) {
    if (th >= th_c) {
      const double d_rad = r_koz - pos_mag;
      const double d_tan = pos_mag * sin(fmin(th - th_c, pi_2));
      out = -1 * fmin(d_rad, d_tan);
    } else {
      out = pos_mag * sin(th_c - th);
    }
  } else {
    if (th >= th_c) {
      out = pos_mag - r_koz;
    } else {
      const double d_rad = pos_mag - r_koz * cos(th_c - th);
      const double d_tan = r_koz * sin(th_c - th);
      out = sqrt(d_rad * d_rad + d_tan * d_tan);
    }
  }
  return out;
}

// get_observation (:294-311) with normalize_value (general.py:243): (b-a)*(val-low)/(high-low)+a, a=-1, b=1
__device__ __forceinline__ void observation(const DevParams& P, const Env& e, float* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = (float)(2.0 * (e.rc[i] - P.obs_lo_r) / P.obs_span_r + -1.0);
#pragma unroll
  for (int i = 0; i < 3; ++i) o[3 + i] = (float)(2.0 * (e.vc[i] - P.obs_lo_v) / P.obs_span_v + -1.0);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[6 + i] = (float)e.qc[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) o[10 + i] = (float)(2.0 * (e.wc[i] - P.obs_lo_w) / P.obs_span_w + -1.0);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[13 + i] = (float)e.qt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter = (env id lo, env id hi, episode, block), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  }
}
__device__ __forceinline__ double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// general.py:248-254: uniform(-1,1,3) normalised (cube-normalised direction, as the reference)
__device__ __forceinline__ void unit_vector(double u0, double u1, double u2, double* o) {
  const double v[3] = {-1.0 + 2.0 * u0, -1.0 + 2.0 * u1, -1.0 + 2.0 * u2};
  const double n = norm3(v);
  o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}
// quaternions.py:11-27
__device__ __forceinline__ void rot2quat(const double* axis_in, double theta, double* q) {
  const double an = norm3(axis_in);
  double s, c;
  sincos(theta / 2, &s, &c);
  q[0] = c; q[1] = axis_in[0] / an * s; q[2] = axis_in[1] / an * s; q[3] = axis_in[2] / an * s;
  const double mag = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= mag; q[1] /= mag; q[2] /= mag; q[3] /= mag;
}
// quaternions.py:149-170 (inputs normalised, output not)
__device__ __forceinline__ void quat_product(const double* a_in, const double* b_in, double* o) {
  const double ma = sqrt(a_in[0] * a_in[0] + a_in[1] * a_in[1] + a_in[2] * a_in[2] + a_in[3] * a_in[3]);
  const double mb = sqrt(b_in[0] * b_in[0] + b_in[1] * b_in[1] + b_in[2] * b_in[2] + b_in[3] * b_in[3]);
  const double a[4] = {a_in[0] / ma, a_in[1] / ma, a_in[2] / ma, a_in[3] / ma};
  const double b[4] = {b_in[0] / mb, b_in[1] / mb, b_in[2] / mb, b_in[3] / mb};
  o[0] = a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
  o[1] = a[0] * b[1] + b[0] * a[1] + (a[2] * b[3] - a[3] * b[2]);
  o[2] = a[0] * b[2] + b[0] * a[2] + (a[3] * b[1] - a[1] * b[3]);
  o[3] = a[0] * b[3] + b[0] * a[3] + (a[1] * b[2] - a[2] * b[1]);
}

// reset (:223-270).  Draw order is the reference's: unit vector then magnitude for rc, vc, wc, wt; angle then axis for qc, qt.
template <typename ST>
__device__ __forceinline__ void reset_env(const DevParams& P, Env& e, uint64_t seed, uint64_t env_id, const double* tape_row) {
  if (tape_row) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.rc[i] = tape_row[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) e.vc[i] = tape_row[3 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) e.qc[i] = tape_row[6 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) e.wc[i] = tape_row[10 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) e.qt[i] = tape_row[13 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) e.wt[i] = tape_row[17 + i];
  } else {
    double u[24];
#pragma unroll
    for (uint32_t j = 0; j < 6; ++j) {
      uint32_t c0 = (uint32_t)env_id, c1 = (uint32_t)(env_id >> 32), c2 = e.episode, c3 = j;
      philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
      u[4 * j + 0] = u01(c0); u[4 * j + 1] = u01(c1); u[4 * j + 2] = u01(c2); u[4 * j + 3] = u01(c3);
    }
    double dir[3], qdev[4], tmp[3], R[9];
    unit_vector(u[0], u[1], u[2], dir);                                   // :231
    { const double m = P.rc0_range * u[3];
      for (int i = 0; i < 3; ++i) e.rc[i] = P.nominal_rc0[i] + dir[i] * m; }   // :253
    unit_vector(u[4], u[5], u[6], dir);                                   // :234
    { const double m = P.vc0_range * u[7];
      for (int i = 0; i < 3; ++i) e.vc[i] = P.nominal_vc0[i] + dir[i] * m; }   // :254
    const double theta_c = P.qc0_range * u[8];                            // :237
    unit_vector(u[9], u[10], u[11], dir);                                 // :238
    rot2quat(dir, theta_c, qdev);                                         // :239
    quat_product(qdev, P.nominal_qc0, e.qc);                              // :255
    unit_vector(u[12], u[13], u[14], dir);                                // :242
    { const double m = P.wc0_range * u[15];
      for (int i = 0; i < 3; ++i) tmp[i] = P.nominal_wc0[i] + dir[i] * m; }
    quat2mat(e.qc, R); matTvec(R, tmp, e.wc);                             // :256 lvlh2chaser
    const double theta_t = P.qt0_range * u[16];                           // :245
    unit_vector(u[17], u[18], u[19], dir);                                // :246
    rot2quat(dir, theta_t, qdev);                                         // :247
    quat_product(qdev, P.nominal_qt0, e.qt);                              // :257
    unit_vector(u[20], u[21], u[22], dir);                                // :250
    { const double m = P.wt0_range * u[23];
      for (int i = 0; i < 3; ++i) tmp[i] = P.nominal_wt0[i] + dir[i] * m; }
    quat2mat(e.qt, R); matTvec(R, tmp, e.wt);                             // :258 lvlh2target
  }
  const ST tag = ST(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) { e.rc[i] = canon(e.rc[i], tag); e.vc[i] = canon(e.vc[i], tag); e.wc[i] = canon(e.wc[i], tag); e.wt[i] = canon(e.wt[i], tag); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { e.qc[i] = canon(e.qc[i], tag); e.qt[i] = canon(e.qt[i], tag); }
  Derived d; double corr_l[3];
  derive(P, e, d, corr_l);
  const bool coll = in_koz(P, d);                                         // :261
  const bool succ = !coll && errors_ok(P, d);                             // :262
  e.flags = (coll ? FLAG_COLLIDED : 0u) | ((succ ? 1u : 0u) << SUCCESS_SHIFT);
  e.bubble = canon(P.bubble_radius0, tag);                                // :263
  e.sum_dv = 0.0; e.sum_dw = 0.0;                                         // :264-265
  e.k = 0;                                                                // :266
  e.ep_ret = 0.0;
  e.episode += 1;
}

struct StepResult {
  float obs[17];
  float reward;
  int done;      // 0/1
  int reason;    // 0 none, 1 obs, 2 time, 3 bubble, 4 attitude
};

// step (:160-221) on one env; `a` are the raw float32 actions (not clipped, as the reference :170).
template <typename ST>
__device__ __forceinline__ void step_env(const DevParams& P, Env& e, const float* a, StepResult& r, Derived& d, double* corr_l) {
  const ST tag = ST(0);
  // :172 delta_v = R(qc) * (a[0:3] * max_delta_v); the product is float32 (float32 array * Python float)
  double Rc0[9];
  quat2mat(e.qc, Rc0);
  const double dvb[3] = {(double)mul_f32_rn(a[0], P.max_delta_v_f32), (double)mul_f32_rn(a[1], P.max_delta_v_f32), (double)mul_f32_rn(a[2], P.max_delta_v_f32)};
  double dv_l[3];
  matvec(Rc0, dvb, dv_l);
  const double vx = e.vc[0] + dv_l[0], vy = e.vc[1] + dv_l[1], vz = e.vc[2] + dv_l[2];   // :176
  const double x = e.rc[0], y = e.rc[1], z = e.rc[2];
  // :177 closed-form Clohessy-Wiltshire propagation (dynamics.py:40-51)
  e.rc[0] = canon(P.phi_xx * x + P.phi_xvx * vx + P.phi_xvy * vy, tag);
  e.rc[1] = canon(P.phi_yx * x + y + P.phi_yvx * vx + P.phi_yvy * vy, tag);
  e.rc[2] = canon(P.phi_zz * z + P.phi_zvz * vz, tag);
  e.vc[0] = canon(P.phi_vxx * x + P.phi_vxvx * vx + P.phi_vxvy * vy, tag);
  e.vc[1] = canon(P.phi_vyx * x + P.phi_vyvx * vx + P.phi_vyvy * vy, tag);
  e.vc[2] = canon(P.phi_vzz * z + P.phi_vzvz * vz, tag);
  // :173, :180 delta_w = a[3:] * max_delta_w is a float64 product (max_delta_w is np.float64)
#pragma unroll
  for (int i = 0; i < 3; ++i) e.wc[i] = e.wc[i] + (double)a[3 + i] * P.max_delta_w;
  integrate_attitude(e.qc, e.wc, P.half_dt, P.dt);   // :181
  integrate_attitude(e.qt, e.wt, P.half_dt, P.dt);   // :184
#pragma unroll
  for (int i = 0; i < 3; ++i) e.wc[i] = canon(e.wc[i], tag);
#pragma unroll
  for (int i = 0; i < 4; ++i) { e.qc[i] = canon(e.qc[i], tag); e.qt[i] = canon(e.qt[i], tag); }

  derive(P, e, d, corr_l);
  const bool inst_coll = in_koz(P, d);
  // :187-190
  if (!(e.flags & FLAG_COLLIDED)) {
    if (inst_coll) e.flags |= FLAG_COLLIDED;
    else if (errors_ok(P, d)) e.flags += (1u << SUCCESS_SHIFT);
  }
  e.k += 1;                                                           // :193 t = round(k*dt, 3)
  const double t = rint((double)e.k * P.dt * 1e3) / 1e3;
  double b = e.bubble - P.bubble_decrease_rate;                       // :196-198
  if (b < P.bubble_min) b = P.bubble_min;
  e.bubble = canon(b, tag);
  // :201-202; float32 sums, see the promotion table in oracle/rdv_oracle.c
  const float sum_v = (fabsf(a[0]) + fabsf(a[1])) + fabsf(a[2]);
  const float sum_w = (fabsf(a[3]) + fabsf(a[4])) + fabsf(a[5]);
  e.sum_dv = (double)((float)e.sum_dv + mul_f32_rn(sum_v, P.max_delta_v_f32));
  e.sum_dw = canon(e.sum_dw + (double)sum_w * P.max_delta_w, tag);

  observation(P, e, r.obs);                                           // :205
  // :355-386
  bool outside = false;
#pragma unroll
  for (int i = 0; i < 17; ++i) outside |= !(r.obs[i] >= -1.0f && r.obs[i] <= 1.0f);   // Box.contains; NaN -> outside
  const bool c_time = t >= P.t_max, c_bubble = d.dist > e.bubble, c_att = d.att > P.max_attitude_error;
  r.done = (outside | c_time | c_bubble | c_att) ? 1 : 0;
  r.reason = outside ? 1 : (c_time ? 2 : (c_bubble ? 3 : (c_att ? 4 : 0)));            // :381 first true
  // :313-353
  double rew = P.att_term * (1 - d.att / P.max_attitude_error);                        // :329
  rew += (double)(mul_f32_rn(P.fuel_scale_f32, sum_v) / P.fuel_div_f32);                        // :333
  if (inst_coll) rew -= P.coll_term;                                                   // :336-337
  if (d.dist < P.koz_radius && !(e.flags & FLAG_COLLIDED)) {                           // :340
    if (d.pos < P.max_rd_error) {                                                      // :348-351
      rew += P.bonus_term * (2 - d.pos / P.max_rd_error);
      if (d.att < P.max_qd_error) rew += P.bonus_term * (2 - d.att / P.max_qd_error);
    }
  }
  e.ep_ret = canon(e.ep_ret + rew, tag);
  r.reward = (float)rew;
}

// diagnostics row (RDV_DIAG_DIM = 8)
__device__ __forceinline__ void diagnostics(const DevParams& P, const Env& e, const Derived& d, double* out) {
  const bool inst = in_koz(P, d);
  out[0] = d.pos; out[1] = d.vel; out[2] = d.att; out[3] = d.rot;
  out[4] = inst ? 1.0 : 0.0;
  out[5] = (!(e.flags & FLAG_COLLIDED) && errors_ok(P, d)) ? 1.0 : 0.0;    // check_success (:406-422)
  out[6] = dist_from_koz(P, d);
  out[7] = (e.flags & FLAG_COLLIDED) ? 1.0 : 0.0;
}

}  // namespace rdv
