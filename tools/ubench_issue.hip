// (write-only outputs are declared "+v": a dead "=v" output lets the compiler send all 8 copies to ONE register, and back-to-back
// writes to the same register cost a lone wave ~3 extra cycles each — itself a finding: see XOR32D_SAMEDST)
// Micro-benchmark (diagnostic): what ONE vector instruction of each kind costs a SIMD, measured as throughput with 1, 2 and 4 waves
// per SIMD issuing independent copies of it (8 register sets, no dependency stalls).  The weights of tools/isa_histogram.py.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

#define KERNEL(NAME, BODY)                                                                                               \
  __global__ void k_##NAME(double* out, unsigned long long* cyc, double a, double b, int iters) {                        \
    double x[8], y[8]; float f[8], g[8]; unsigned u[8], w[8]; unsigned long long q[8];                                   \
    for (int c = 0; c < 8; ++c) { x[c] = a + threadIdx.x * 1e-9 + c; y[c] = b + c; f[c] = (float)x[c]; g[c] = 1.0f + c;     \
                                  u[c] = threadIdx.x * 2654435761u + c; w[c] = u[c] ^ 0x9e3779b9u; q[c] = u[c]; }           \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                      \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                          \
    for (int i = 0; i < iters; ++i) {                                                                                    \
      _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) { REP8(BODY) }                                                 \
    }                                                                                                                    \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                          \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                      \
    double s = 0; for (int c = 0; c < 8; ++c) s += x[c] + y[c] + f[c] + g[c] + u[c] + w[c] + (double)q[c];               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                      \
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }                           \
  }

#define B_FMA64(c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
#define B_MUL64(c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
#define B_ADD64(c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
#define B_FMA64S(c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "s"(b), "v"(a));
#define B_RSQ64(c) asm volatile("v_rsq_f64 %0, %1" : "+v"(y[c]) : "v"(x[c]));
#define B_RCP64(c) asm volatile("v_rcp_f64 %0, %1" : "+v"(y[c]) : "v"(x[c]));
#define B_SQRT64(c) asm volatile("v_sqrt_f64 %0, %1" : "+v"(y[c]) : "v"(x[c]));
#define B_RNDNE64(c) asm volatile("v_rndne_f64 %0, %1" : "+v"(y[c]) : "v"(x[c]));
#define B_CVT_F32_F64(c) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f[c]) : "v"(x[c]));
#define B_CVT_F64_F32(c) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(y[c]) : "v"(f[c]));
#define B_CVT_I32_F64(c) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(u[c]) : "v"(x[c]));
#define B_CVT_F64_U32(c) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(y[c]) : "v"(u[c]));
#define B_CMP64(c) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(x[c]), "v"(y[c]) : "vcc");
#define B_MAX64(c) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
#define B_FMA32(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(g[c]), "v"(g[c]));
#define B_ADD32(c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(g[c]));
#define B_PKFMA32(c) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(y[c]));
#define B_MOV32(c) asm volatile("v_mov_b32 %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_MOV64(c) asm volatile("v_mov_b64 %0, %1" : "+v"(y[c]) : "v"(x[c]));
#define B_CNDMASK(c) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(u[c]) : "v"(w[c]), "v"(g[c]));
#define B_MOV32E64(c) asm volatile("v_mov_b32_e64 %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_MOV32LIT(c) asm volatile("v_mov_b32 %0, 0x3ff80000" : "+v"(u[c]));
#define B_XOR32D(c) asm volatile("v_xor_b32 %0, %1, %2" : "+v"(u[c]) : "v"(w[c]), "v"(g[c]));
#define B_XOR32D_SAMEDST(c) asm volatile("v_xor_b32 v40, %0, %1" : : "v"(w[c]), "v"(g[c]) : "v40");
#define B_FMAC64(c) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
#define B_FMA64D(c) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(y[c]) : "v"(x[c]), "v"(b), "v"(a));
#define B_CMP32(c) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(f[c]), "v"(g[c]) : "vcc");
#define B_SMOV(c) asm volatile("s_mov_b32 s20, 0x3ff80000" : : : "s20");
#define B_SADD(c) asm volatile("s_add_u32 s20, s20, 3" : : : "s20", "scc");
#define B_SAND64(c) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");
#define B_SMOV_D(c) asm volatile("s_mov_b32 s2" #c ", 0x3ff80000" : : : "s2" #c);
#define PAIR_0 "s[40:41]"
#define PAIR_1 "s[42:43]"
#define PAIR_2 "s[44:45]"
#define PAIR_3 "s[46:47]"
#define PAIR_4 "s[48:49]"
#define PAIR_5 "s[50:51]"
#define PAIR_6 "s[52:53]"
#define PAIR_7 "s[54:55]"
#define SCLOB "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define B_SAND64_D(c) asm volatile("s_and_b64 " PAIR_##c ", s[56:57], exec" : : : SCLOB);
#define B_SAVEEXEC(c) asm volatile("s_and_saveexec_b64 " PAIR_##c ", vcc\n s_or_b64 exec, exec, " PAIR_##c : : : SCLOB);
#define B_CMP32_D(c) asm volatile("v_cmp_le_f32 " PAIR_##c ", %0, %1" : : "v"(f[c]), "v"(g[c]) : SCLOB);
#define B_VALU_SALU(c) asm volatile("v_add_f32 %0, %0, %1\n s_mov_b32 s2" #c ", 0x3ff80000" : "+v"(f[c]) : "v"(g[c]) : "s2" #c);
#define B_SNOP(c) asm volatile("s_nop 0");
// a Horner step both ways: constant moved into the accumulator (what the compiler emits) vs constant as an SGPR-pair operand
#define B_HORNER_MOV(c) asm volatile("v_mov_b32 %0, 0x55555555\n v_mov_b32 %1, 0x3fa55555\n v_fmac_f64 %2, %3, %4" : "+v"(u[c]), "+v"(w[c]), "+v"(y[c]) : "v"(x[c]), "v"(b));
#define B_HORNER_SGPR(c) asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x3fa55555\n v_fma_f64 %0, %1, %2, s[20:21]" : "+v"(y[c]) : "v"(x[c]), "v"(b) : "s20", "s21");
#define B_HORNER_SLOADED(c) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(y[c]) : "v"(x[c]), "v"(b), "s"(a));
#define B_XOR32(c) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_ADDU32(c) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_MULLO(c) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_MULHI(c) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
#define B_MAD64(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[c]) : "v"(u[c]), "v"(w[c]) : "vcc");
#define B_LSHLADD64(c) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[c]) : "v"(x[c]));
#define B_RCP32(c) asm volatile("v_rcp_f32 %0, %1" : "+v"(g[c]) : "v"(f[c]));
#define B_EXP32(c) asm volatile("v_exp_f32 %0, %1" : "+v"(g[c]) : "v"(f[c]));
#define B_DPP(c) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(w[c]));
#define B_READLANE(c) asm volatile("v_readlane_b32 s20, %0, 63" : : "v"(w[c]) : "s20");
#define B_CVTPK(c) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "+v"(u[c]) : "v"(f[c]));

#define ALL(X) X(FMA64) X(MUL64) X(ADD64) X(FMA64S) X(RSQ64) X(RCP64) X(SQRT64) X(RNDNE64) X(CVT_F32_F64) X(CVT_F64_F32) X(CVT_I32_F64) \
  X(CVT_F64_U32) X(CMP64) X(MAX64) X(FMA32) X(ADD32) X(PKFMA32) X(MOV32) X(MOV64) X(CNDMASK) X(XOR32) X(ADDU32) X(MULLO) X(MULHI) X(MAD64) \
  X(LSHLADD64) X(RCP32) X(EXP32) X(DPP) X(READLANE) X(CVTPK) \
  X(MOV32E64) X(MOV32LIT) X(XOR32D) X(XOR32D_SAMEDST) X(FMAC64) X(FMA64D) X(CMP32) X(SMOV) X(SADD) X(SAND64) X(SMOV_D) X(SAND64_D) X(SAVEEXEC) X(CMP32_D) X(VALU_SALU) X(SNOP) X(HORNER_MOV) X(HORNER_SGPR) X(HORNER_SLOADED)
#define DEF(N) KERNEL(N, B_##N)
ALL(DEF)

typedef void (*kern_t)(double*, unsigned long long*, double, double, int);
struct Entry { const char* name; kern_t k; };
#define ENT(N) {#N, k_##N},
static Entry entries[] = {ALL(ENT)};

int main() {
  const int iters = 2000, per_iter = 32;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 4096 * 64); hipMalloc(&cyc, 16 * 4096);
  printf("%-14s %9s %9s %9s   (shader cycles per instruction per SIMD; 1 / 2 / 4 waves per SIMD; clock GHz)\n", "instruction", "1 wave", "2 waves", "4 waves");
  for (auto& e : entries) {
    double res[3]; double ghz = 0;
    int wi = 0;
    for (int waves : {1, 2, 4}) {
      const int blocks = 1024 * waves;   // 64-thread blocks: one wave each, dealt over the 1024 SIMDs
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(2 * blocks);
      hipMemcpy(h.data(), cyc, 16 * blocks, hipMemcpyDeviceToHost);
      double m = 0, r = 0; for (int i = 0; i < blocks; ++i) { m += h[2 * i]; r += h[2 * i + 1]; }
      res[wi++] = m / blocks / ((double)iters * per_iter) / waves;
      ghz = m / (r * 10.0);
    }
    printf("%-14s %9.2f %9.2f %9.2f   %.2f\n", e.name, res[0], res[1], res[2], ghz);
  }
  return 0;
}
