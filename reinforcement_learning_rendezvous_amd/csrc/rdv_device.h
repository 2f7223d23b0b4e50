// rdv_device.h — per-lane device math of the fused rendezvous step for gfx950 (CDNA4, wave64).
//
// One lane owns one environment.  All arithmetic is fp64 (MI355X vector fp64 runs at half the fp32 rate and the path
// needs ~1 kFLOP per env-step, so precision is free relative to the launch; it removes the fp32 hazards of SURVEY §7:
// the 1e-5 rounding of the cosines in general.py:179 and the cancellation in the CW matrix).  The persistent state is
// held in HBM in the storage type ST (float in production, double in parity mode) and is canonicalised to ST right
// after propagation, so that everything derived in the same step (flags, observation, reward, diagnostics) is a
// function of exactly the stored state.
//
// The kernel is latency-bound (one wave per SIMD at N = 65,536), so the instruction stream is kept short:
//   - no fp64 division or sqrt on the step path: normalisations use v_rsq_f64 + 2 Newton steps, every constant divisor
//     is a host-precomputed reciprocal;
//   - the attitude update needs cos(t) and sin(t)/t only, both even in t: two 10-term polynomials in t^2 = (|w|dt/2)^2,
//     no sqrt, no range reduction (|w|dt/2 <= pi/4 always holds inside the observation Box; a general path remains);
//   - the rounded-cosine angle tests of general.py:179 (acos(round(c,5)) vs a fixed angle) are integer comparisons of
//     k = rint(c*1e5) against thresholds the host derives with the same libm acos the oracle uses — exact, and no acos;
//     acos is evaluated once per step, for the attitude term of the reward.
// Results differ from the straight-line oracle by a few fp64 ulps (tests/test_gpu_parity.py states the tolerances).
//
// Citations are file:line in cfdeinza/reinforcement-learning-rendezvous.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdv {

// ---------------------------------------------------------------------------------------------------------------
// Parameter block, derived once on the host (derive_params in rdv_hip.hip).  The step kernels read it through a
// top-level __restrict__ pointer to device memory, so every field is a scalar load into SGPRs (wave-uniform).
struct DevParams {
  // Clohessy-Wiltshire state-transition matrix for (n, dt), the 14 non-zeros of dynamics.py:40-47
  double phi_xx, phi_xvx, phi_xvy;      // row 0: 4-3c, s/n, 2(1-c)/n
  double phi_yx, phi_yvx, phi_yvy;      // row 1: 6(s-nt), -2(1-c)/n, (4s-3nt)/n     (phi_yy = 1)
  double phi_zz, phi_zvz;               // row 2: c, s/n
  double phi_vxx, phi_vxvx, phi_vxvy;   // row 3: 3ns, c, 2s
  double phi_vyx, phi_vyvx, phi_vyvy;   // row 4: -6n(1-c), -2s, 4c-3
  double phi_vzz, phi_vzvz;             // row 5: -ns, c
  double dt, half_dt;
  double max_delta_w;                   // np.float64 in the reference (:82) -> fp64 product (NEP 50)
  float  max_delta_v_f32;               // Python float * float32 array stays float32 (:172, :201)
  float  fuel_scale_f32;                // float32(dt * fuel_coef)  (:333)
  float  fuel_div_f32;                  // float32(3 * max_delta_v) (:333)
  int32_t k_time;                       // first step count k with round(k*dt, 3) >= t_max (:193, :368)
  double obs_lo_r, obs_span_r, obs_inv_span_r;   // normalize_value (general.py:243): low, high-low, RN(1/(high-low))
  double obs_lo_v, obs_span_v, obs_inv_span_v;
  double obs_lo_w, obs_span_w, obs_inv_span_w;
  double koz_radius, corridor_half_angle;
  double inv_max_attitude_error, inv_max_rd_error, inv_max_qd_error;
  double corridor_axis[3], capture_axis[3], rd[3];
  double inv_corridor_norm, inv_capture_norm;
  // The reference compares NORMS with limits: np.linalg.norm(x) <= limit (:416-417), < limit (:348, :397).  The kernel keeps the
  // sum of squares s (formed as the reference's NumPy forms it: sumsq3) and compares it with
  // the largest double T for which sqrt(T) <= limit (le2_*) or sqrt(T) < limit (lt2_*), found on the host with the correctly rounded
  // sqrt: since sqrt is monotone, s <= T decides exactly what sqrt(s) <= limit decides — no sqrt, and no off-by-an-ulp of limit^2.
  double le2_rd, lt2_rd, le2_vd, lt2_vd, le2_wd, lt2_wd, lt2_koz;
  // thresholds on k = rint(1e5*cos) equivalent to the reference's tests on acos(round(cos,5)) (general.py:179)
  double reset_flag_radius2;            // (max(koz_radius, |rd| + max_rd_error))^2: beyond it a fresh state has collided = success = 0
  double kc_coll_max;                   // in corridor cone test (:401): angle > half_angle      <=> k <= kc_coll_max
  double ka_done_max;                   // :370 attitude error > max_attitude_error               <=> k <= ka_done_max
  double ka_succ_min;                   // :417 attitude error <= max_qd_error                    <=> k >= ka_succ_min
  double ka_bonus_min;                  // :350 attitude error <  max_qd_error                    <=> k >= ka_bonus_min
  double bubble_radius0, bubble_decrease_rate, bubble_min;
  double att_term, coll_term, bonus_term;   // dt*att_coef, dt*collision_coef, dt*bonus_coef
  const double* acos_table;                 // device: acos(k/1e5), k = -100000..100000, filled by the host's libm
  // reset (rendezvous_env.py:229-258); nominal quaternions pre-normalised (quat_product does it, quaternions.py:159-160)
  double nominal_rc0[3], nominal_vc0[3], nominal_qc0[4], nominal_wc0[3], nominal_qt0[4], nominal_wt0[3];
  double rc0_range, vc0_range, qc0_range, wc0_range, qt0_range, wt0_range;
  int32_t qc0_tiny, qt0_tiny;             // (range / 2)^2 <= kTinyU: a reset's attitude deviation uses the short series (deviate)
  // general rigid bodies (rdv_set_rigid_body; read by the kGeneral kernels only): inertia tensor, its inverse (row-major) and the
  // constant body torque of the chaser [0] and the target [1]; tolerances of the env's solve_ivp calls (:567-568)
  double body_inertia[2][9], body_inv_inertia[2][9], body_torque[2][3];
  double rk_rtol, rk_atol;
  int32_t body_general[2];                // per body (chaser, target): integrate with RK45 (else the closed form applies to it)
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

// What the flags / reward / done / diagnostics need from the (canonical) state.
struct Derived {
  double r2;                // |rc|^2, NumPy's sum of squares
  double k_att, k_corr;     // rint(1e5 * cos) of the attitude error (:432) and of the corridor angle (:400)
  double pos2, vel2, rot2;  // squared get_errors() entries (:463-466), NumPy's sums of squares
};

__device__ __forceinline__ double canon(double x, float) { return (double)(float)x; }
__device__ __forceinline__ double canon(double x, double) { return x; }

// float32 multiply that is never contracted into an fma (NumPy rounds the product, then the sum)
__device__ __forceinline__ float mul_f32_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// 1/sqrt(x): v_rsq_f64 seed + two Newton steps (<= 2 ulp); replaces the sqrt-then-divide of every normalisation
__device__ __forceinline__ double rsqrt64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

__device__ __forceinline__ double dot3(const double* a, const double* b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
// np.linalg.norm(v)**2 before the square root: v.dot(v) as the BLAS of the reference's NumPy forms it for three elements — this
// same ascending chain of fused multiply-adds (oracle/rdv_oracle.c, norm3; pinned by tests/golden/boundary_diag.npz)
__device__ __forceinline__ double sumsq3(const double* v) { return dot3(v, v); }

// cos(t) and sin(t)/t from u = t*t, for u <= kSmallU (t <= ~pi/4): Taylor to u^9, truncation < 2e-18
constexpr double kSmallU = 0.62;
__device__ __forceinline__ void cos_sinc_small(double u, double& c, double& sc) {
  double p = -1.0 / 6402373705728000.0;            // -1/18!
  p = fma(p, u, 1.0 / 20922789888000.0);           //  1/16!
  p = fma(p, u, -1.0 / 87178291200.0);             // -1/14!
  p = fma(p, u, 1.0 / 479001600.0);                //  1/12!
  p = fma(p, u, -1.0 / 3628800.0);                 // -1/10!
  p = fma(p, u, 1.0 / 40320.0);                    //  1/8!
  p = fma(p, u, -1.0 / 720.0);                     // -1/6!
  p = fma(p, u, 1.0 / 24.0);                       //  1/4!
  p = fma(p, u, -0.5);
  c = fma(p, u, 1.0);
  double q = -1.0 / 121645100408832000.0;          // -1/19!
  q = fma(q, u, 1.0 / 355687428096000.0);          //  1/17!
  q = fma(q, u, -1.0 / 1307674368000.0);           // -1/15!
  q = fma(q, u, 1.0 / 6227020800.0);               //  1/13!
  q = fma(q, u, -1.0 / 39916800.0);                // -1/11!
  q = fma(q, u, 1.0 / 362880.0);                   //  1/9!
  q = fma(q, u, -1.0 / 5040.0);                    // -1/7!
  q = fma(q, u, 1.0 / 120.0);                      //  1/5!
  q = fma(q, u, -1.0 / 6.0);
  sc = fma(q, u, 1.0);
}
// The same two series cut after u^4, for u <= kTinyU = 2^-7 (t <= 0.088 rad = 5 deg): the first dropped terms are u^5/10! < 8e-18 and
// u^5/11! < 8e-19, a twentieth of an ulp of results near 1.  Every attitude step inside the observation Box is in this range
// (|w| dt/2 <= 10 deg/s * dt/2), as is a reset's chaser attitude: 8 Horner steps instead of 18 (and as many 64-bit literals less).
constexpr double kTinyU = 0.0078125;
// (The eight coefficients as SGPR operands read from the parameter block — one s_load instead of sixteen v_mov — were measured in round 4:
//  24 instructions fewer per call and 0.09 us SLOWER per launch at 65,536 envs; the scalar load's latency lands on the step wave's chain.
//  profiles/r04_isa_diet.txt)
__device__ __forceinline__ void cos_sinc_tiny(double u, double& c, double& sc) {
  double p = 1.0 / 40320.0;                        //  1/8!
  p = fma(p, u, -1.0 / 720.0);                     // -1/6!
  p = fma(p, u, 1.0 / 24.0);                       //  1/4!
  p = fma(p, u, -0.5);
  c = fma(p, u, 1.0);
  double q = 1.0 / 362880.0;                       //  1/9!
  q = fma(q, u, -1.0 / 5040.0);                    // -1/7!
  q = fma(q, u, 1.0 / 120.0);                      //  1/5!
  q = fma(q, u, -1.0 / 6.0);
  sc = fma(q, u, 1.0);
}
// Larger angles (never reached inside the observation Box: |w| <= 10 deg/s): halve the angle until it is small,
// then apply cos 2x = 2cos^2 x - 1, sinc 2x = sinc x cos x once per halving (error doubles per halving).
__device__ __forceinline__ void cos_sinc_large(double u, double& c, double& sc) {
  int halvings = 0;
#pragma clang loop unroll(disable)
  while (__builtin_expect(u > kSmallU && halvings < 64, 0)) { u *= 0.25; ++halvings; }
  cos_sinc_small(u, c, sc);
#pragma clang loop unroll(disable)
  for (int h = 0; h < halvings; ++h) { sc = sc * c; c = fma(2.0 * c, c, -1.0); }
}
// Which series a lane uses is a function of ITS u alone (a per-lane branch; never of what the other lanes of its wave hold), so an env
// gets the same bits from every kernel, whatever that kernel's grouping of envs into waves.
__device__ __forceinline__ void cos_sinc(double u, double& c, double& sc) {
  if (__builtin_expect(u <= kTinyU, 1)) cos_sinc_tiny(u, c, sc);
  else cos_sinc_large(u, c, sc);
}

// R(q) of quaternions.py:48-68 (which normalises q first, :57)
__device__ __forceinline__ void quat2mat(const double* q, double* m) {
  const double inv = rsqrt64(fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0]))));
  const double qw = q[0] * inv, qx = q[1] * inv, qy = q[2] * inv, qz = q[3] * inv;
  const double ww = qw * qw;
  m[0] = fma(2.0, fma(qx, qx, ww), -1.0); m[1] = 2.0 * fma(qx, qy, -qw * qz);      m[2] = 2.0 * fma(qx, qz, qw * qy);
  m[3] = 2.0 * fma(qx, qy, qw * qz);      m[4] = fma(2.0, fma(qy, qy, ww), -1.0);  m[5] = 2.0 * fma(qy, qz, -qw * qx);
  m[6] = 2.0 * fma(qx, qz, -qw * qy);     m[7] = 2.0 * fma(qy, qz, qw * qx);       m[8] = fma(2.0, fma(qz, qz, ww), -1.0);
}
__device__ __forceinline__ void matvec(const double* m, const double* v, double* o) {
  o[0] = fma(m[2], v[2], fma(m[1], v[1], m[0] * v[0]));
  o[1] = fma(m[5], v[2], fma(m[4], v[1], m[3] * v[0]));
  o[2] = fma(m[8], v[2], fma(m[7], v[1], m[6] * v[0]));
}
__device__ __forceinline__ void matTvec(const double* m, const double* v, double* o) {
  o[0] = fma(m[6], v[2], fma(m[3], v[1], m[0] * v[0]));
  o[1] = fma(m[7], v[2], fma(m[4], v[1], m[1] * v[0]));
  o[2] = fma(m[8], v[2], fma(m[5], v[1], m[2] * v[0]));
}

// rendezvous_env.py:552-604 with the reference's isotropic inertia and zero torque: w is constant and
// q(t+dt) = normalize(q (x) [cos(|w|dt/2), w_hat sin(|w|dt/2)])  (body-frame rate: right multiplication, dynamics.py:137-151)
// The reference's right-hand side normalises q before differentiating it (dynamics.py:109, :134) while the integrated state is the
// quaternion as given: for |q| = rho the solution keeps its norm and turns at w / rho.  Every step ends with a normalisation, so
// rho != 1 only in the first step after a state was injected from outside (rdv_set_state): kRaw builds handle it, the kernels of
// every other step do not pay for the test.
template <bool kRaw = false>
__device__ __forceinline__ void integrate_attitude(double* q, const double* w, double half_dt) {
  double u = dot3(w, w) * (half_dt * half_dt);           // (|w| dt/2)^2
  if (kRaw) {
    const double n2 = fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0])));
    if (fabs(n2 - 1.0) > 1e-6) {
      half_dt *= rsqrt64(n2);
      u = (dot3(w, w) * half_dt) * half_dt;              // in this order: a body at rest stays exact for any |q| > 0
    }
  }
  double c, sc;
  cos_sinc(u, c, sc);
  const double k = sc * half_dt;                          // sin(|w|dt/2)/|w|
  const double dx = w[0] * k, dy = w[1] * k, dz = w[2] * k;
  const double a = q[0], b = q[1], cc = q[2], d = q[3];
  const double o0 = fma(a, c, -fma(b, dx, fma(cc, dy, d * dz)));
  const double o1 = fma(a, dx, fma(b, c, fma(cc, dz, -d * dy)));
  const double o2 = fma(a, dy, fma(cc, c, fma(d, dx, -b * dz)));
  const double o3 = fma(a, dz, fma(b, dy, fma(d, c, -cc * dx)));
  const double inv = rsqrt64(fma(o3, o3, fma(o2, o2, fma(o1, o1, o0 * o0))));   // :574, :601
  q[0] = o0 * inv; q[1] = o1 * inv; q[2] = o2 * inv; q[3] = o3 * inv;
}

// ---------------------------------------------------------------------------------------------------------------
// General rigid body (SURVEY f-4): any inertia tensor, constant body torque.  The reference integrates
//   q' = 1/2 Omega(w) q/|q|,  w' = I^-1 (tau - w x I w)           (dynamics.py:93-175)
// with scipy.integrate.solve_ivp(method='RK45', rtol=1e-7, atol=1e-6) over one dt (rendezvous_env.py:561-570, :588-597).
// This is the same Dormand-Prince 5(4) pair with scipy's initial-step rule and step-size controller, operation for operation
// (fused multiply-adds off in this section), in fp64: it takes the same accepted / rejected steps as scipy and lands within
// rounding of its result.  Each lane runs its own adaptive loop (divergent trip counts; typically 1-3 steps per dt).
// The right-hand side normalises the quaternion twice (dynamics.py:109 derivative_of_att_and_rot_rate, then :134 quat_derivative):
// q / |q| each time.  Here each is ONE reciprocal square root (v_rsq_f64 + two Newton steps, <= 2 ulp: rsqrt64) and four products
// instead of a square root and four divisions: 24 instructions instead of ~136 of a right-hand side of ~160, evaluated 7 times per
// attempted step (round 4: tri-axial target 28 -> see profiles/r04_rigid_time.txt).  The quotients differ from the reference's by an
// ulp or two; the step sequence (accept / reject, step sizes) is scipy's unless an error norm lies within ~1e-15 of 1.
__device__ __forceinline__ void rigid_rhs(const double* I, const double* Iinv, const double* tau, const double* y, double* dy) {
#pragma clang fp contract(off)
  const double inv1 = rsqrt64(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);       // dynamics.py:109
  const double a0 = y[0] * inv1, a1 = y[1] * inv1, a2 = y[2] * inv1, a3 = y[3] * inv1;
  const double inv2 = rsqrt64(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);                       // dynamics.py:134
  const double q0 = a0 * inv2, q1 = a1 * inv2, q2 = a2 * inv2, q3 = a3 * inv2;
  const double w1 = y[4], w2 = y[5], w3 = y[6];
  dy[0] = 0.5 * (-w1 * q1 - w2 * q2 - w3 * q3);                                              // dynamics.py:137-151
  dy[1] = 0.5 * (w1 * q0 + w3 * q2 - w2 * q3);
  dy[2] = 0.5 * (w2 * q0 - w3 * q1 + w1 * q3);
  dy[3] = 0.5 * (w3 * q0 + w2 * q1 - w1 * q2);
  double L[3], r[3];                                                                         // dynamics.py:169-171
#pragma unroll
  for (int i = 0; i < 3; ++i) L[i] = I[3 * i] * w1 + I[3 * i + 1] * w2 + I[3 * i + 2] * w3;
  r[0] = tau[0] - (w2 * L[2] - w3 * L[1]);
  r[1] = tau[1] - (w3 * L[0] - w1 * L[2]);
  r[2] = tau[2] - (w1 * L[1] - w2 * L[0]);
#pragma unroll
  for (int i = 0; i < 3; ++i) dy[4 + i] = Iinv[3 * i] * r[0] + Iinv[3 * i + 1] * r[1] + Iinv[3 * i + 2] * r[2];
}

__device__ __forceinline__ double rms7(const double* x) {
#pragma clang fp contract(off)
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) s += x[i] * x[i];
  return sqrt(s) / sqrt(7.0);
}

// x^(-1/5) for x > 0 — the step-size controller's 0.9 * error_norm^(-1/5) (scipy rk.py) and the initial step's x^(1/5) = x * (x^(-1/5))^4.
// libm's pow is ~230 instructions a call, and a wave whose lanes split over "accepted" and "rejected" runs both calls of an attempt:
// the controller cost a third of an attempted step.  Here: x = m 2^e with e = 5k + j, so x^(-1/5) = 2^(-k) (m 2^j)^(-1/5) with
// t = m 2^j in [0.5, 16); a float32 estimate of t^(-1/5) from the hardware's log2 / exp2 (relative error ~1e-6) and two Newton steps
// y <- y (6 - t y^5) / 5 in fp64 (error e -> 3 e^2: 1e-6 -> 3e-12 -> below an ulp).  Within a few ulp of pow; ~35 instructions.
__device__ __forceinline__ double pow_minus_fifth(double x) {
  if (!(x < 1e300)) return 0.0;                      // huge / inf / NaN: the callers clamp the factor from below (0.2) anyway
  int e;
  const double m = frexp(x, &e);                     // x = m 2^e, m in [0.5, 1)
  const int k = (e >= 0 ? e : e - 4) / 5;            // floor(e / 5)
  const double t = ldexp(m, e - 5 * k);              // [0.5, 16)
  double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)t));
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double y2 = y * y;
    y = (y * fma(-t, y2 * y2 * y, 6.0)) * 0.2;
  }
  return ldexp(y, -k);
}

constexpr int kRk45MaxAttempts = 4096;   // a lane that has not finished dt by then gets a NaN state (scipy would be failing too)

__device__ __forceinline__ void integrate_attitude_rk45(double* q, double* w, const double* I, const double* Iinv, const double* tau,
                                                        double t_bound, double rtol, double atol) {
#pragma clang fp contract(off)
  // Dormand-Prince coefficients (scipy/integrate/_ivp/rk.py RK45: A, B, E)
  constexpr double A[6][5] = {{0, 0, 0, 0, 0},
                              {1.0 / 5, 0, 0, 0, 0},
                              {3.0 / 40, 9.0 / 40, 0, 0, 0},
                              {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                              {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                              {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
  constexpr double B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
  constexpr double E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
  double y[7] = {q[0], q[1], q[2], q[3], w[0], w[1], w[2]};
  double f[7];
  rigid_rhs(I, Iinv, tau, y, f);
  double h_abs;
  {  // select_initial_step (common.py), error order 4
    double sa[7], sb[7], scale[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { scale[i] = atol + fabs(y[i]) * rtol; sa[i] = y[i] / scale[i]; sb[i] = f[i] / scale[i]; }
    const double d0 = rms7(sa), d1 = rms7(sb);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    if (h0 > t_bound) h0 = t_bound;
    double y1[7], f1[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) y1[i] = y[i] + h0 * f[i];
    rigid_rhs(I, Iinv, tau, y1, f1);
#pragma unroll
    for (int i = 0; i < 7; ++i) sa[i] = (f1[i] - f[i]) / scale[i];
    const double d2 = rms7(sa) / h0;
    double h1 = fmax(1e-6, h0 * 1e-3);
    if (!(d1 <= 1e-15 && d2 <= 1e-15)) {             // (0.01 / max(d1, d2))^(1/5) = x * (x^(-1/5))^4
      const double x = 0.01 / fmax(d1, d2), r = pow_minus_fifth(x), r2 = r * r;
      h1 = x * (r2 * r2);
    }
    h_abs = fmin(fmin(100 * h0, h1), t_bound);
  }
  double t = 0.0;
  int attempts = 0;
  bool rejected = false;
  bool failed = false;
  double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
  if (h_abs < min_step) h_abs = min_step;
#pragma clang loop unroll(disable)
  while (t < t_bound) {   // one ATTEMPT per trip (accepted or rejected); every lane leaves after at most kRk45MaxAttempts trips
    double h = h_abs, t_new = t + h;
    if (t_new - t_bound > 0) t_new = t_bound;
    h = t_new - t;
    h_abs = fabs(h);
    double K[7][7], tmp[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) K[0][i] = f[i];
#pragma unroll
    for (int s = 1; s < 6; ++s) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        double dy = 0.0;
#pragma unroll
        for (int j = 0; j < s; ++j) dy += K[j][i] * A[s][j];
        tmp[i] = y[i] + dy * h;
      }
      rigid_rhs(I, Iinv, tau, tmp, K[s]);
    }
    double y_new[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < 6; ++s) acc += K[s][i] * B[s];
      y_new[i] = y[i] + h * acc;
    }
    rigid_rhs(I, Iinv, tau, y_new, K[6]);
    double err[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < 7; ++s) acc += K[s][i] * E[s];
      const double sc = atol + fmax(fabs(y[i]), fabs(y_new[i])) * rtol;
      err[i] = acc * h / sc;
    }
    const double error_norm = rms7(err);
    ++attempts;
    if (error_norm < 1) {
      double factor = error_norm < 1e-8 ? 10.0 : fmin(10.0, 0.9 * pow_minus_fifth(error_norm));   // (below 1e-8 the power exceeds 10 / 0.9)
      if (rejected && factor > 1) factor = 1;
      h_abs *= factor;
      t = t_new;
#pragma unroll
      for (int i = 0; i < 7; ++i) { y[i] = y_new[i]; f[i] = K[6][i]; }
      rejected = false;
      min_step = 10 * fabs(nextafter(t, INFINITY) - t);
      if (h_abs < min_step) h_abs = min_step;
    } else {
      if (!(error_norm == error_norm) || h_abs <= min_step) { failed = true; break; }   // non-finite state / step size too small
      h_abs *= fmax(0.2, 0.9 * pow_minus_fifth(error_norm));
      if (h_abs < min_step) h_abs = min_step;
      rejected = true;
    }
    if (attempts >= kRk45MaxAttempts) { failed = t < t_bound; break; }
  }
  if (failed) {
#pragma unroll
    for (int i = 0; i < 7; ++i) y[i] = __builtin_nan("");
  }
  const double mag = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);   // :574, :601
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = y[i] / mag;
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = y[4 + i];
}

// Flags/reward inputs from the canonical state: one R(qc), one R(qt) per call (the reference rebuilds them ~10x per step).
// Two halves, so that the step can finish the chaser side first and have the attitude error's table entry (attitude_error_of)
// travelling while the target side computes:
//   derive_chaser : R(qc), |rc|, k_att                      derive_target : R(qt), position error, and the lazy terms
// kLazy (training kernels): the velocity / rotation-rate errors are only ever compared after the position error has
// passed its limit (:417 np.all, :348), and the corridor angle only inside the KOZ sphere (:397); far from the target
// they are skipped and set to +inf, which fails exactly the comparisons the reference would fail.
__device__ __forceinline__ void derive_chaser(const DevParams& P, const Env& e, Derived& d, double& inv_dist) {
  double Rc[9];
  quat2mat(e.qc, Rc);
  double cap_l[3];
  matvec(Rc, P.capture_axis, cap_l);    // :431
  const double r2 = sumsq3(e.rc);
  inv_dist = rsqrt64(r2);
  d.r2 = r2;
  // general.py:179: round(cos, 5) == rint(cos*1e5)/1e5; rotations preserve |capture_axis|, |corridor_axis|
  d.k_att = rint(-dot3(e.rc, cap_l) * (inv_dist * P.inv_capture_norm) * 1e5);           // :432
}
// (R(qc) is rebuilt inside the one branch that needs it, from the same canonical quaternion — the same matrix, bit for bit: carried
//  over from derive_chaser it occupied 18 registers across the target's attitude step and the whole of this function, which is where
//  the fused kernels' register pressure peaked: tools/vgpr_pressure.py.  The quaternion is passed through an empty asm there so that
//  the compiler does not recognise the expression as the one derive_chaser evaluated and keep that result alive instead.)
template <bool kLazy>
__device__ __forceinline__ void derive_target(const DevParams& P, const Env& e, Derived& d, double inv_dist) {
  double Rt[9];
  quat2mat(e.qt, Rt);
  double rd_l[3];
  matvec(Rt, P.rd, rd_l);               // :460
  const double dp[3] = {e.rc[0] - rd_l[0], e.rc[1] - rd_l[1], e.rc[2] - rd_l[2]};
  d.pos2 = sumsq3(dp);                  // :463
  const double inf = __builtin_huge_val();
  d.vel2 = inf; d.rot2 = inf; d.k_corr = inf;
  // the corridor angle first: after it only the next branch needs R(qt), and that one is done with it before it builds R(qc)
  if (!kLazy || d.r2 <= P.lt2_koz) {
    double corr_l[3];
    matvec(Rt, P.corridor_axis, corr_l);                                                // :400
    d.k_corr = rint(dot3(e.rc, corr_l) * (inv_dist * P.inv_corridor_norm) * 1e5);
  }
  if (!kLazy || d.pos2 <= P.le2_rd) {
    double wt_l[3];
    matvec(Rt, e.wt, wt_l);             // :459
    const double vd_l[3] = {fma(wt_l[1], rd_l[2], -wt_l[2] * rd_l[1]), fma(wt_l[2], rd_l[0], -wt_l[0] * rd_l[2]),
                            fma(wt_l[0], rd_l[1], -wt_l[1] * rd_l[0])};                 // :461
    const double dv[3] = {e.vc[0] - vd_l[0], e.vc[1] - vd_l[1], e.vc[2] - vd_l[2]};
    d.vel2 = sumsq3(dv);                // :464
    // R(qc) only now (the asm also takes the target-side results as inputs, which orders it behind them: one matrix alive at a time)
    double Rc[9], wc_l[3];
    double qc_again[4] = {e.qc[0], e.qc[1], e.qc[2], e.qc[3]};
    asm volatile("" : "+v"(qc_again[0]), "+v"(qc_again[1]), "+v"(qc_again[2]), "+v"(qc_again[3]), "+v"(wt_l[0]), "+v"(wt_l[1]), "+v"(wt_l[2]), "+v"(d.vel2));
    quat2mat(qc_again, Rc);
    matvec(Rc, e.wc, wc_l);             // :458
    const double dw[3] = {wc_l[0] - wt_l[0], wc_l[1] - wt_l[1], wc_l[2] - wt_l[2]};
    d.rot2 = sumsq3(dw);                // :466
  }
}
template <bool kLazy>
__device__ __forceinline__ void derive(const DevParams& P, const Env& e, Derived& d) {
  double inv_dist;
  derive_chaser(P, e, d, inv_dist);
  derive_target<kLazy>(P, e, d, inv_dist);
}

// k / 1e5 correctly rounded (k is an integer-valued double, |k| <= 1e5): product by 1e-5 plus one residual correction
__device__ __forceinline__ double div_1e5(double k) {
  const double q = k * 1e-5;
  return fma(fma(-q, 1e5, k), 1e-5, q);
}
__device__ __forceinline__ double angle_of(double k) { return acos(div_1e5(k)); }     // general.py:179
// The same value on the step path: a rounded cosine takes one of 200,001 values k*1e-5, so acos(k/1e5) is read from a
// 1.6 MB table the host fills with its libm acos (the oracle's), instead of ~110 instructions of device acos.  Only the
// ~14k entries of attitude errors below 30 deg are ever touched (episodes end beyond): ~110 KB, L2-resident.
__device__ __forceinline__ double attitude_error_of(const DevParams& P, double k) {
  const bool valid = fabs(k) <= 100000.0;                    // false for NaN (rc = 0: the reference asserts there)
  const int idx = valid ? (int)k + 100000 : 100000;
  const double a = P.acos_table[idx];
  return valid ? a : __builtin_nan("");
}

// check_collision (:388-404)
__device__ __forceinline__ bool in_koz(const DevParams& P, const Derived& d) {
  return d.r2 <= P.lt2_koz && d.k_corr <= P.kc_coll_max;       // norm(rc) < koz_radius (:397)
}
// np.all(errors <= error_ranges) (:416-417)
__device__ __forceinline__ bool errors_ok(const DevParams& P, const Derived& d) {
  return d.pos2 <= P.le2_rd && d.vel2 <= P.le2_vd && d.k_att >= P.ka_succ_min && d.rot2 <= P.le2_wd;
}

// dist_from_koz (:510-537) — evaluator-only
__device__ __forceinline__ double dist_from_koz(const DevParams& P, const Derived& d) {
  const double pos_mag = sqrt(d.r2), r_koz = P.koz_radius, th_c = P.corridor_half_angle;   // np.linalg.norm (:519)
  const double th = angle_of(d.k_corr);
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

// normalize_value (general.py:243): (b-a)*(val-low)/(high-low)+a with a=-1, b=1.  The quotient is the correctly rounded
// y/span without a division: q = y*RN(1/span), one fma for the exact residual, one to apply it (Markstein), so the
// float32 observation is bit-identical to the reference's for an identical state.
__device__ __forceinline__ float normalized(double val, double lo, double span, double inv_span) {
#pragma clang fp contract(off)
  const double y = 2.0 * (val - lo);
  double q = y * inv_span;
  q = fma(fma(-q, span, y), inv_span, q);
  return (float)(q + -1.0);
}
// get_observation (:294-311).  The 17 floats are handed to `sink(j, value)` one by one as they are formed — a kernel whose sink writes
// the env's staged row in LDS never holds the observation in registers (as an array it was live from here to the row store, across
// done / reward / statistics: 17 of the step kernels' registers).
template <typename Sink>
__device__ __forceinline__ void observation_to(const DevParams& P, const Env& e, Sink&& sink) {
#pragma unroll
  for (int i = 0; i < 3; ++i) sink(i, normalized(e.rc[i], P.obs_lo_r, P.obs_span_r, P.obs_inv_span_r));
#pragma unroll
  for (int i = 0; i < 3; ++i) sink(3 + i, normalized(e.vc[i], P.obs_lo_v, P.obs_span_v, P.obs_inv_span_v));
#pragma unroll
  for (int i = 0; i < 4; ++i) sink(6 + i, (float)e.qc[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) sink(10 + i, normalized(e.wc[i], P.obs_lo_w, P.obs_span_w, P.obs_inv_span_w));
#pragma unroll
  for (int i = 0; i < 4; ++i) sink(13 + i, (float)e.qt[i]);
}
__device__ __forceinline__ void observation(const DevParams& P, const Env& e, float* o) {
  observation_to(P, e, [&](int j, float v) { o[j] = v; });
}
// the sink of a row of floats in memory (LDS staging rows, slots): element j of the observation -> row[j]
struct RowSink {
  float* row;
  __device__ __forceinline__ void operator()(int j, float v) const { row[j] = v; }
};

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter = (env id lo, env id hi, episode, block), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    // one full 32x32->64 product each (v_mad_u64_u32) instead of separate quarter-rate mul_hi + mul_lo
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  }
}
// 21-bit uniform in (0,1).  A reset needs 24 uniforms = 504 bits = 4 Philox blocks (three 21-bit fields per 64 bits);
// 32-bit integer multiplies are quarter-rate on CDNA, so the block count is what a reset costs.
// ((field + 0.5) / 2^21 as ONE fma: field * 2^-21 + 2^-22 — every intermediate of either form is exact, so the same bits)
__device__ __forceinline__ double u21(uint32_t field) { return fma((double)field, 1.0 / 2097152.0, 0.5 / 2097152.0); }

// The same draw mapped onto (-1, 1): 2 u - 1 = field * 2^-20 + (2^-21 - 1) as ONE fma — exact like u21 and like fma(2, u21, -1): the same bits.
// Eighteen of a reset's 24 uniforms are only ever used in that form (the three components of its six random directions).
__device__ __forceinline__ double s21(uint32_t field) { return fma((double)field, 1.0 / 1048576.0, 1.0 / 2097152.0 - 1.0); }
// which of the 24 uniforms of a reset are direction components (u[0..2] rc, u[4..6] vc, u[9..11] qc axis, u[12..14] wc, u[17..19] qt axis, u[20..22] wt)
constexpr uint32_t kSignedUniforms = (7u << 0) | (7u << 4) | (7u << 9) | (7u << 12) | (7u << 17) | (7u << 20);

// general.py:248-254: uniform(-1,1,3) normalised (cube-normalised direction, as the reference); v = the three draws already on (-1, 1)
__device__ __forceinline__ void unit_vector(double v0, double v1, double v2, double* o) {
  const double v[3] = {v0, v1, v2};
  const double inv = rsqrt64(dot3(v, v));
  o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}
// quat_product(rot2quat(axis, theta), nominal) (:239/:255, quaternions.py:11-27, :149-170).  `axis` is a unit vector and
// `nominal` is pre-normalised, so the reference's re-normalisations of axis, of the rotation quaternion and of both
// factors are identities up to rounding and are dropped; as in the reference, the product itself is not normalised.
// `tiny`: wave-uniform, from the PARAMETERS (0 <= theta <= the sampling range, and (range/2)^2 <= kTinyU) — so that the series does
// not depend on what a wave happens to hold, and a wave of mostly large angles (the target's 45 deg range) does not run both.
__device__ __forceinline__ void deviate(const double* axis, double theta, const double* nominal, bool tiny, double* o) {
  const double half = 0.5 * theta;
  double c, sc;
  if (tiny) cos_sinc_tiny(half * half, c, sc);
  else cos_sinc_large(half * half, c, sc);
  const double s = sc * half;
  const double a[4] = {c, axis[0] * s, axis[1] * s, axis[2] * s};
  const double* b = nominal;
  o[0] = fma(a[0], b[0], -fma(a[1], b[1], fma(a[2], b[2], a[3] * b[3])));
  o[1] = fma(a[0], b[1], fma(b[0], a[1], fma(a[2], b[3], -a[3] * b[2])));
  o[2] = fma(a[0], b[2], fma(b[0], a[2], fma(a[3], b[1], -a[1] * b[3])));
  o[3] = fma(a[0], b[3], fma(b[0], a[3], fma(a[1], b[2], -a[2] * b[1])));
}

// One Philox block of the reset stream -> its six 21-bit uniforms u[6j .. 6j+5], each in the form its one user wants: on (-1, 1) for the
// direction components (kSignedUniforms), on (0, 1) for the magnitudes and angles
__device__ __forceinline__ void reset_uniforms(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t j, double* u) {
  uint32_t c0 = (uint32_t)env_id, c1 = (uint32_t)(env_id >> 32), c2 = episode, c3 = j;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t f[6] = {c0 & 0x1FFFFFu, ((c0 >> 21) | (c1 << 11)) & 0x1FFFFFu, (c1 >> 10) & 0x1FFFFFu,
                         c2 & 0x1FFFFFu, ((c2 >> 21) | (c3 << 11)) & 0x1FFFFFu, (c3 >> 10) & 0x1FFFFFu};
#pragma unroll
  for (int k = 0; k < 6; ++k) u[6 * j + k] = ((kSignedUniforms >> (6 * j + k)) & 1u) ? s21(f[k]) : u21(f[k]);   // (j is a literal at every call site)
}

// Which part of a new initial state a call computes.  The six sampled quantities of reset() are independent given their
// uniforms (only wc needs R(qc) and wt needs R(qt), :256, :258), so the preparation of a next-episode state can be shared out
// over several waves, each running a short instruction stream over the same list of envs (prepared-state slots,
// csrc/rdv_slots.h): the expressions — hence the results, bit for bit — are those of the whole reset.
enum ResetPart : int { RESET_ALL = -1, RESET_RC_VC = 0, RESET_QC_WC = 1, RESET_QT = 2, RESET_WT = 3 };

// reset (:223-262): the new state (the fields of kPart; the others are left untouched) from (seed, env id, e.episode) or from a
// tape row.  Draw order is the reference's: unit vector then magnitude for rc, vc, wc, wt; angle then axis for qc, qt.
// kCanonAll = false (the split kernel's service waves, which compute every env's next state speculatively): only rc is rounded to the
// storage type here — what the flags test below needs; the other 17 values are rounded where a finished env actually takes the state
// (canon_rest), which is the same rounding applied later: 34 conversions less in front of the kernel's barrier for ~95 % of the lanes.
template <typename ST, int kPart, bool kCanonAll = true>
__device__ __forceinline__ void reset_fields(const DevParams& P, Env& e, uint64_t seed, uint64_t env_id, const double* tape_row) {
  constexpr bool all = kPart == RESET_ALL;
  constexpr bool do_rv = all || kPart == RESET_RC_VC, do_c = all || kPart == RESET_QC_WC, do_qt = all || kPart == RESET_QT || kPart == RESET_WT,
                 do_wt = all || kPart == RESET_WT;
  if (tape_row) {
    if (do_rv) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { e.rc[i] = tape_row[i]; e.vc[i] = tape_row[3 + i]; }
    }
    if (do_c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e.qc[i] = tape_row[6 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) e.wc[i] = tape_row[10 + i];
    }
    if (do_qt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e.qt[i] = tape_row[13 + i];
    }
    if (do_wt) {
#pragma unroll
      for (int i = 0; i < 3; ++i) e.wt[i] = tape_row[17 + i];
    }
  } else {
    // u[0..3] rc, u[4..7] vc, u[8..11] qc, u[12..15] wc, u[16..19] qt, u[20..23] wt; block j holds u[6j .. 6j+5]
    double u[24];
    if (do_rv) reset_uniforms(seed, env_id, e.episode, 0, u);
    if (do_rv || do_c) reset_uniforms(seed, env_id, e.episode, 1, u);
    if (do_c || do_qt) reset_uniforms(seed, env_id, e.episode, 2, u);
    if (do_qt) reset_uniforms(seed, env_id, e.episode, 3, u);
    double dir[3], tmp[3], R[9];
    if (do_rv) {
      unit_vector(u[0], u[1], u[2], dir);                                   // :231
      { const double m = P.rc0_range * u[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) e.rc[i] = fma(dir[i], m, P.nominal_rc0[i]); }   // :253
      unit_vector(u[4], u[5], u[6], dir);                                   // :234
      { const double m = P.vc0_range * u[7];
#pragma unroll
        for (int i = 0; i < 3; ++i) e.vc[i] = fma(dir[i], m, P.nominal_vc0[i]); }   // :254
    }
    if (do_c) {
      const double theta_c = P.qc0_range * u[8];                            // :237
      unit_vector(u[9], u[10], u[11], dir);                                 // :238
      deviate(dir, theta_c, P.nominal_qc0, P.qc0_tiny != 0, e.qc);                           // :239, :255
      unit_vector(u[12], u[13], u[14], dir);                                // :242
      { const double m = P.wc0_range * u[15];
#pragma unroll
        for (int i = 0; i < 3; ++i) tmp[i] = fma(dir[i], m, P.nominal_wc0[i]); }
      quat2mat(e.qc, R); matTvec(R, tmp, e.wc);                             // :256 lvlh2chaser
    }
    if (do_qt) {
      const double theta_t = P.qt0_range * u[16];                           // :245
      unit_vector(u[17], u[18], u[19], dir);                                // :246
      deviate(dir, theta_t, P.nominal_qt0, P.qt0_tiny != 0, e.qt);                           // :247, :257
    }
    if (do_wt) {
      unit_vector(u[20], u[21], u[22], dir);                                // :250
      { const double m = P.wt0_range * u[23];
#pragma unroll
        for (int i = 0; i < 3; ++i) tmp[i] = fma(dir[i], m, P.nominal_wt0[i]); }
      quat2mat(e.qt, R); matTvec(R, tmp, e.wt);                             // :258 lvlh2target
    }
  }
  const ST tag = ST(0);
  if (do_rv) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.rc[i] = canon(e.rc[i], tag);
  }
  if (!kCanonAll) return;
  if (do_rv) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.vc[i] = canon(e.vc[i], tag);
  }
  if (do_c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.wc[i] = canon(e.wc[i], tag);
#pragma unroll
    for (int i = 0; i < 4; ++i) e.qc[i] = canon(e.qc[i], tag);
  }
  if (do_qt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) e.qt[i] = canon(e.qt[i], tag);
  }
  if (do_wt) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.wt[i] = canon(e.wt[i], tag);
  }
}

// :261-262 collided = check_collision(), success = int(check_success()) of a complete new state.  Both need the chaser within
// max(koz_radius, |rd| + max_rd_error) of the target (inside the KOZ sphere, resp. position error <= max_rd_error);
// the nominal start is 10 m out, so the rotation matrices and errors are only built for states that close.
__device__ __forceinline__ bool reset_flags_needed(const DevParams& P, const Env& e) { return dot3(e.rc, e.rc) < P.reset_flag_radius2; }
__device__ __forceinline__ uint32_t reset_flags(const DevParams& P, const Env& e) {
  Derived d;
  derive<true>(P, e, d);
  const bool coll = in_koz(P, d);
  const bool succ = !coll && errors_ok(P, d);
  return (coll ? FLAG_COLLIDED : 0u) | ((succ ? 1u : 0u) << SUCCESS_SHIFT);
}

// the rounding reset_fields<.., kCanonAll = false> left out (everything but rc)
template <typename ST>
__device__ __forceinline__ void canon_rest(Env& e) {
  const ST tag = ST(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) { e.vc[i] = canon(e.vc[i], tag); e.wc[i] = canon(e.wc[i], tag); e.wt[i] = canon(e.wt[i], tag); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { e.qc[i] = canon(e.qc[i], tag); e.qt[i] = canon(e.qt[i], tag); }
}

template <typename ST, bool kCanonAll = true>
__device__ __forceinline__ void reset_state(const DevParams& P, Env& e, uint64_t seed, uint64_t env_id, const double* tape_row) {
  reset_fields<ST, RESET_ALL, kCanonAll>(P, e, seed, env_id, tape_row);
  e.flags = 0u;
  if (reset_flags_needed(P, e)) {            // (never with the reference's nominal start 10 m out)
    if (!kCanonAll) canon_rest<ST>(e);       // the flags are a function of the stored state (idempotent: the taker rounds again)
    e.flags = reset_flags(P, e);
  }
}

// the bookkeeping half of reset (:263-266)
template <typename ST>
__device__ __forceinline__ void reset_aux(const DevParams& P, Env& e) {
  e.bubble = canon(P.bubble_radius0, ST(0));                              // :263
  e.sum_dv = 0.0; e.sum_dw = 0.0;                                         // :264-265
  e.k = 0;                                                                // :266
  e.ep_ret = 0.0;
  e.episode += 1;
}

template <typename ST>
__device__ __forceinline__ void reset_env(const DevParams& P, Env& e, uint64_t seed, uint64_t env_id, const double* tape_row) {
  reset_state<ST>(P, e, seed, env_id, tape_row);
  reset_aux<ST>(P, e);
}

struct StepResult {   // (the observation goes to the caller's sink: see observation_to)
  float reward;
  double reward64;   // the reward before its float32 store (the reference's evaluators sum the float64 value, monte_carlo.py:150)
  int done;      // 0/1
  int reason;    // 0 none, 1 obs, 2 time, 3 bubble, 4 attitude
};

// step (:160-221) on one env; `a` are the raw float32 actions (not clipped, as the reference :170).
// `mid()` is called once the chaser side is finished and the attitude error's table entry has been requested — the last load of a
// transition: a kernel that fetches ahead (step_kernel_tiles) issues its look-ahead loads there, BEHIND that entry in the in-order
// vector-memory counter, so that waiting for the entry does not mean waiting for the look-ahead.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
  __device__ __forceinline__ void operator()(double&) const {}
};

// What the first half of a transition hands to the second (step_env = step_env_chaser + step_target + step_env_finish).
struct StepCtx {
  float sum_v, sum_w;      // :201-202, :333: the action enters the bookkeeping and the reward only through these two float32 sums
  double inv_dist, att;    // 1/|rc| and the attitude error (the table entry requested at the end of the chaser half)
};

// First half (:172-181): impulse, Clohessy-Wiltshire propagation, the chaser's rate impulse and attitude step, then the chaser's half
// of the derived quantities.  kGeneral: a body whose inertia tensor / torque is not the reference's (rdv_set_rigid_body) is integrated
// with the reference's own scheme (RK45, both its attitude and its rate evolve) — PER BODY (P.body_general[0 | 1], wave-uniform): a
// tumbling tri-axial target beside the reference's chaser leaves the chaser on the closed form.
// `pre()` is called once, behind the last thing that does not need the action (the chaser's rotation matrix) and in front of the first
// use of `a`: a kernel whose action row is still in flight waits for it there (step_kernel_split / _parts: PinnedInputs).
template <typename ST, bool kGeneral = false, bool kRaw = false, typename Pre = NoHook>
__device__ __forceinline__ void step_env_chaser(const DevParams& P, Env& e, const float* a, Derived& d, StepCtx& c, Pre&& pre = Pre()) {
  const ST tag = ST(0);
  // :172 delta_v = R(qc) * (a[0:3] * max_delta_v); the product is float32 (float32 array * Python float)
  double Rc0[9];
  quat2mat(e.qc, Rc0);
  pre(Rc0[8]);   // (handed the matrix's last element: a wait written as inline assembly takes it as an operand, so it cannot be scheduled in front of the matrix)
  // formed here, so that the six action registers die with the impulses below instead of living to the end of the transition
  c.sum_v = (fabsf(a[0]) + fabsf(a[1])) + fabsf(a[2]);
  c.sum_w = (fabsf(a[3]) + fabsf(a[4])) + fabsf(a[5]);
  const double dvb[3] = {(double)mul_f32_rn(a[0], P.max_delta_v_f32), (double)mul_f32_rn(a[1], P.max_delta_v_f32),
                         (double)mul_f32_rn(a[2], P.max_delta_v_f32)};
  double dv_l[3];
  matvec(Rc0, dvb, dv_l);
  const double vx = e.vc[0] + dv_l[0], vy = e.vc[1] + dv_l[1], vz = e.vc[2] + dv_l[2];   // :176
  const double x = e.rc[0], y = e.rc[1], z = e.rc[2];
  // :177 closed-form Clohessy-Wiltshire propagation (dynamics.py:40-51), summed in the reference's column order
  e.rc[0] = canon(fma(P.phi_xvy, vy, fma(P.phi_xvx, vx, P.phi_xx * x)), tag);
  e.rc[1] = canon(fma(P.phi_yvy, vy, fma(P.phi_yvx, vx, fma(P.phi_yx, x, y))), tag);
  e.rc[2] = canon(fma(P.phi_zvz, vz, P.phi_zz * z), tag);
  e.vc[0] = canon(fma(P.phi_vxvy, vy, fma(P.phi_vxvx, vx, P.phi_vxx * x)), tag);
  e.vc[1] = canon(fma(P.phi_vyvy, vy, fma(P.phi_vyvx, vx, P.phi_vyx * x)), tag);
  e.vc[2] = canon(fma(P.phi_vzvz, vz, P.phi_vzz * z), tag);
  // :173, :180 delta_w = a[3:] * max_delta_w is a float64 product (max_delta_w is np.float64)
#pragma unroll
  for (int i = 0; i < 3; ++i) e.wc[i] = fma((double)a[3 + i], P.max_delta_w, e.wc[i]);
  // Chaser side first (:181), then its half of the derived quantities: the attitude error's table entry (the one dependent memory
  // access of a transition) is requested here and is needed ~500 instructions later, for the reward.
  if (kGeneral && P.body_general[0]) integrate_attitude_rk45(e.qc, e.wc, P.body_inertia[0], P.body_inv_inertia[0], P.body_torque[0], P.dt, P.rk_rtol, P.rk_atol);
  else integrate_attitude<kRaw || kGeneral>(e.qc, e.wc, P.half_dt);   // (the general kernels also run the first step after rdv_set_state)
#pragma unroll
  for (int i = 0; i < 3; ++i) e.wc[i] = canon(e.wc[i], tag);
#pragma unroll
  for (int i = 0; i < 4; ++i) e.qc[i] = canon(e.qc[i], tag);
  derive_chaser(P, e, d, c.inv_dist);
  c.att = attitude_error_of(P, d.k_att);
}

// The target's attitude step (:184) on (qt, wt) alone: what a partner wave can run beside the chaser half (step_kernel_general).
// The result is NOT yet canonical: step_env_finish rounds it to the storage type.
template <bool kGeneral = false, bool kRaw = false>
__device__ __forceinline__ void step_target(const DevParams& P, double* qt, double* wt) {
  if (kGeneral && P.body_general[1]) integrate_attitude_rk45(qt, wt, P.body_inertia[1], P.body_inv_inertia[1], P.body_torque[1], P.dt, P.rk_rtol, P.rk_atol);
  else integrate_attitude<kRaw || kGeneral>(qt, wt, P.half_dt);
}

// Second half (:187-221): the target's state made canonical, its half of the derived quantities, the latches, bookkeeping, reward,
// observation and termination.
template <typename ST, bool kLazy, bool kGeneral = false, typename Sink>
__device__ __forceinline__ void step_env_finish(const DevParams& P, Env& e, StepResult& r, Derived& d, const StepCtx& c, Sink&& sink) {
  const ST tag = ST(0);
  if (kGeneral) {
#pragma unroll
    for (int i = 0; i < 3; ++i) e.wt[i] = canon(e.wt[i], tag);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) e.qt[i] = canon(e.qt[i], tag);
  derive_target<kLazy>(P, e, d, c.inv_dist);
  const bool inst_coll = in_koz(P, d);
  // :187-190
  if (!(e.flags & FLAG_COLLIDED)) {
    if (inst_coll) e.flags |= FLAG_COLLIDED;
    else if (errors_ok(P, d)) e.flags += (1u << SUCCESS_SHIFT);
  }
  e.k += 1;                                                           // :193 t = round(k*dt, 3)
  e.bubble = canon(fmax(e.bubble - P.bubble_decrease_rate, P.bubble_min), tag);   // :196-198
  // :201-202; float32 sums, see the promotion table in oracle/rdv_oracle.c
  e.sum_dv = (double)((float)e.sum_dv + mul_f32_rn(c.sum_v, P.max_delta_v_f32));
  e.sum_dw = canon(fma((double)c.sum_w, P.max_delta_w, e.sum_dw), tag);

  // :313-353 (the reward does not depend on the observation or on done: computed first, the observation last, so that its 17 floats
  // go straight to the sink)
  const double att = c.att;
  double rew = P.att_term * fma(-att, P.inv_max_attitude_error, 1.0);                  // :329
  rew += (double)(mul_f32_rn(P.fuel_scale_f32, c.sum_v) / P.fuel_div_f32);             // :333
  if (inst_coll) rew -= P.coll_term;                                                   // :336-337
  if (d.r2 <= P.lt2_koz && !(e.flags & FLAG_COLLIDED) && d.pos2 <= P.lt2_rd) {             // :340, :348 (both strict: lt2_*)
    rew += P.bonus_term * fma(-sqrt(d.pos2), P.inv_max_rd_error, 2.0);                 // :349
    if (d.k_att >= P.ka_bonus_min) rew += P.bonus_term * fma(-att, P.inv_max_qd_error, 2.0);   // :350-351
  }
  e.ep_ret = canon(e.ep_ret + rew, tag);
  r.reward = (float)rew;
  r.reward64 = rew;
  // :205, :355-386
  bool outside = false;
  observation_to(P, e, [&](int j, float v) {
    outside |= !(fabsf(v) <= 1.0f);                                                    // Box.contains; NaN -> outside
    sink(j, v);
  });
  // norm(rc) > bubble_radius (:369): the bubble changes every step, so the sum of squares is compared with bubble^2 bracketed by its
  // rounding, and the correctly rounded square root decides inside the bracket (one wave in ~10^14 goes there)
  const double b2 = e.bubble * e.bubble;
  bool c_bubble = d.r2 > b2 * (1.0 + 1e-15);
  if (__builtin_expect(!c_bubble && d.r2 > b2 * (1.0 - 1e-15), 0)) c_bubble = sqrt(d.r2) > e.bubble;
  const bool c_time = e.k >= P.k_time, c_att = d.k_att <= P.ka_done_max;
  r.done = (outside | c_time | c_bubble | c_att) ? 1 : 0;
  r.reason = outside ? 1 : (c_time ? 2 : (c_bubble ? 3 : (c_att ? 4 : 0)));            // :381 first true
}

// step (:160-221) on one env; `a` are the raw float32 actions (not clipped, as the reference :170).
template <typename ST, bool kLazy, bool kGeneral = false, bool kRaw = false, typename Sink, typename Hook = NoHook, typename Pre = NoHook>
__device__ __forceinline__ void step_env(const DevParams& P, Env& e, const float* a, StepResult& r, Derived& d, Sink&& sink, Hook&& mid = Hook(),
                                         Pre&& pre = Pre()) {
  StepCtx c;
  step_env_chaser<ST, kGeneral, kRaw>(P, e, a, d, c, pre);
  mid();
  step_target<kGeneral, kRaw>(P, e.qt, e.wt);      // :184
  step_env_finish<ST, kLazy, kGeneral>(P, e, r, d, c, sink);
}

// diagnostics row (RDV_DIAG_DIM = 8) — evaluator-only
__device__ __forceinline__ void diagnostics(const DevParams& P, const Env& e, const Derived& d, double* out) {
  out[0] = sqrt(d.pos2); out[1] = sqrt(d.vel2); out[2] = angle_of(d.k_att); out[3] = sqrt(d.rot2);
  out[4] = in_koz(P, d) ? 1.0 : 0.0;
  out[5] = (!(e.flags & FLAG_COLLIDED) && errors_ok(P, d)) ? 1.0 : 0.0;    // check_success (:406-422)
  out[6] = dist_from_koz(P, d);
  out[7] = (e.flags & FLAG_COLLIDED) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-env evaluation accumulators (RDV_EVAL_DIM = 32 doubles per env), updated by the evaluator build of the step kernel where an
// env executed a transition: everything the reference's evaluators gather step by step on the host, kept on the device —
//   custom_callbacks.evaluate_policy (:211-267): sum of attitude errors, steps inside the keep-out zone, time of the first one,
//     smallest position error before it, total reward;
//   monte_carlo.evaluate (:117-205): number of collision / success steps, minimum distance from the KOZ, and the terminal errors:
//     the mean of each error from the first step at which all four (else three, else two, else one) constraints hold (:153-189) —
//     kept as running sums per level, which needs no error history: a level's sum starts at its first hit and runs to the end.
// Layout: 0 total reward, 1 steps, 2 sum att, 3 collision steps, 4 t of first collision (NaN: none), 5 min pos error before it (NaN: none),
// 6 success steps, 7 min dist from KOZ, 8-11 last errors, then four levels {count, sum pos, vel, att, rot}: 12 all four, 17 three, 22 two, 27 one.
constexpr int kEvalDim = 32;
// dg: the diagnostics row of the state the accumulators are updated with; first = true for the initial state (k = 0)
__device__ __forceinline__ void eval_accumulate(const DevParams& P, const Env& e, const Derived& d, const double* dg, double reward,
                                                bool first, double* acc) {
  const bool koz = dg[4] != 0.0;
  const double t_now = rint((double)e.k * P.dt * 1e3) / 1e3;                     // :193
  if (first) {
    acc[0] = 0.0; acc[1] = 0.0; acc[2] = dg[2];                                   // :213
    acc[3] = koz ? 1.0 : 0.0;                                                     // :215-220
    acc[4] = koz ? 0.0 : __builtin_nan("");
    acc[5] = koz ? __builtin_nan("") : dg[0];                                     // :221-222
    acc[6] = dg[5]; acc[7] = dg[6];                                               // monte_carlo.py:121-123
  } else {
    acc[0] += reward; acc[1] += 1.0; acc[2] += dg[2];                             // :242-243
    const bool no_coll_yet = acc[4] != acc[4];
    if (koz) { acc[3] += 1.0; if (no_coll_yet) acc[4] = t_now; }                  // :244-248
    else if (no_coll_yet) acc[5] = fmin(acc[5], dg[0]);                           // :249-252
    acc[6] += dg[5];                                                              // monte_carlo.py:145-146
    acc[7] = fmin(acc[7], dg[6]);                                                 // monte_carlo.py:147-149
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[8 + j] = dg[j];
  // monte_carlo.py:159-162: strict comparisons of the four errors with their limits
  const bool p = d.pos2 <= P.lt2_rd, v = d.vel2 <= P.lt2_vd, a = d.k_att >= P.ka_bonus_min, r = d.rot2 <= P.lt2_wd;
  const bool hit[4] = {p && v && a && r, (p && v && a) || (p && v && r), p && v, p};      // :163, :167, :172, :175
#pragma unroll
  for (int L = 0; L < 4; ++L) {
    double* lv = acc + 12 + 5 * L;
    if (first) { lv[0] = 0.0; lv[1] = lv[2] = lv[3] = lv[4] = 0.0; }
    if (lv[0] > 0.0 || hit[L]) {
      lv[0] += 1.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) lv[1 + j] += dg[j];
    }
  }
}

}  // namespace rdv
