// Pre-pass of the streaming (deep) levels: GroupNorm(+FiLM)(+SiLU) over the channel concat of two
// sources, or LayerNorm, applied ONCE to a small channel-last tensor ahead of a weight-streaming GEMM
// (reference jen1/model/blocks.py:140-144, :427, :530, :732-734).
//
// The tensor is tiny (<= 24 x 2048 per batch element), so a launch is one memory round trip plus its
// instruction count.  One thread = one 8-channel vector.  There is no LDS and no barrier: every thread
// merges the fine-group sums of ITS group straight from the producers' statistics (a few broadcast
// loads that are in flight together with the activation vector, gamma / beta and the FiLM rows), and
// the kernel arguments arrive as one explicit batch of scalar loads.
#include "common.h"

namespace {

struct NormDev {
  const void* x0;
  const void* x1;
  void* y;
  const float* st0;
  const float* st1;
  const float* gamma;
  const float* beta;
  const float* film;         // already offset by film_off
  const int32_t* film_row;
  const int32_t* film_step;
  const float* ln_rowstats;
  int32_t mode, L, c0, c1, ld0, ld1, ld_y, groups, cpg, vpr, nvec, film_C, film_ld, nfg0, nfg1, cpf0, cpf1;
  float inv_vpr, inv_cpg, inv_count, eps, src1_scale, inv_cpf0, inv_cpf1;
  int32_t pad[2];
};
static_assert(sizeof(NormDev) == 192, "three s_load_dwordx16");

typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ NormDev load_norm_args() {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  u32x16 k0, k1, k2;
  asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx16 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(k0), "=&s"(k1), "=&s"(k2) : "s"(kp) : "memory");
  struct Raw { unsigned d[48]; } raw;
#pragma unroll
  for (int i = 0; i < 16; ++i) { raw.d[i] = k0[i]; raw.d[16 + i] = k1[i]; raw.d[32 + i] = k2[i]; }
  return __builtin_bit_cast(NormDev, raw);
}

template <typename T>
__global__ __launch_bounds__(256) void norm_apply_kernel(const NormDev a_unused) {
  constexpr bool PRECISE = is_f32<T>::value;
  const NormDev a = load_norm_args();
  const int b = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= a.nvec) return;
  const int row = (int)(((float)v + 0.5f) * a.inv_vpr);          // exact: nvec < 2^20
  const int c = (v - row * a.vpr) * 8;
  const bool gn = (a.mode != JEN1_PRO_LN);
  const bool s1 = c >= a.c0;
  const int grow = b * a.L + row;

  // ---- every load of this thread, back to back ---------------------------------------------------
  float x[8], gam[8], bet[8];
  if (s1) load8(reinterpret_cast<const T*>(a.x1) + (size_t)((unsigned)grow * (unsigned)a.ld1 + (unsigned)(c - a.c0)), x);
  else load8(reinterpret_cast<const T*>(a.x0) + (size_t)((unsigned)grow * (unsigned)a.ld0 + (unsigned)c), x);
  const bool affine = a.gamma != nullptr;
  if (affine) {
    load8(a.gamma + c, gam);
    load8(a.beta + c, bet);
  }
  float fs[8], fh[8];
  const bool film = gn && a.film != nullptr;
  if (film) {
    const int fr = a.film_step ? a.film_step[0] : (a.film_row ? a.film_row[b] : b);
    const float* fp = a.film + (size_t)((unsigned)fr * (unsigned)a.film_ld + (unsigned)c);
    load8(fp, fs);
    load8(fp + a.film_C, fh);
  }
  float mean, rstd;
  if (gn) {
    // group of this vector and the fine-group sums it is made of (consecutive (sum, sumsq) pairs)
    const int gch = (int)(((float)c + 0.5f) * a.inv_cpg);
    const int g = gch < a.groups ? gch : a.groups - 1;
    const int lo = g * a.cpg - (s1 ? a.c0 : 0);
    const int nfg = s1 ? a.nfg1 : a.nfg0;
    const int f0 = (int)(((float)lo + 0.5f) * (s1 ? a.inv_cpf1 : a.inv_cpf0));
    const float2* fine = reinterpret_cast<const float2*>((s1 ? a.st1 : a.st0) + b * 64) + f0;
    float s = 0.f, q = 0.f;
    {
      float2 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = fine[(k < nfg && f0 + k < JEN1_FINE_GROUPS) ? k : 0];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool in = k < nfg && f0 + k < JEN1_FINE_GROUPS;
        s += in ? t[k].x : 0.f;
        q += in ? t[k].y : 0.f;
      }
    }
    for (int k = 8; k < nfg && f0 + k < JEN1_FINE_GROUPS; ++k) {
      s += fine[k].x;
      q += fine[k].y;
    }
    const float sc = s1 ? a.src1_scale : 1.0f;
    s *= sc;
    q *= sc * sc;
    mean = s * a.inv_count;
    float var = q * a.inv_count - mean * mean;
    var = var < 0.f ? 0.f : var;
    rstd = PRECISE ? 1.0f / sqrtf(var + a.eps) : rsqrtf(var + a.eps);
    // y = (x*sc - mean) * rstd * gamma + beta, then FiLM, then SiLU
    float A[8], Bc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      A[j] = rstd * (affine ? gam[j] : 1.0f);
      Bc[j] = (affine ? bet[j] : 0.0f) - mean * A[j];
      A[j] *= sc;
      if (film) {
        const float f1 = fs[j] + 1.0f;
        A[j] *= f1;
        Bc[j] = Bc[j] * f1 + fh[j];
      }
    }
    if ((a.cpg & 7) != 0) {
      // groups narrower than a vector (tiny configurations): per-channel statistics, same formula
      const float* st = s1 ? a.st1 : a.st0;
      const int cpf = s1 ? a.cpf1 : a.cpf0;
#pragma unroll 1
      for (int j = 0; j < 8; ++j) {
        int gj = (c + j) / a.cpg;
        gj = gj < a.groups ? gj : a.groups - 1;
        const int lo_j = gj * a.cpg - (s1 ? a.c0 : 0);
        const int hi_j = ((gj == a.groups - 1) ? (a.c0 + a.c1) : (gj + 1) * a.cpg) - (s1 ? a.c0 : 0);
        float sj = 0.f, qj = 0.f;
        for (int f = lo_j / cpf; f < (hi_j + cpf - 1) / cpf && f < JEN1_FINE_GROUPS; ++f) {
          sj += st[b * 64 + 2 * f];
          qj += st[b * 64 + 2 * f + 1];
        }
        sj *= sc;
        qj *= sc * sc;
        const float mj = sj * a.inv_count;
        float vj = qj * a.inv_count - mj * mj;
        vj = vj < 0.f ? 0.f : vj;
        const float rj = PRECISE ? 1.0f / sqrtf(vj + a.eps) : rsqrtf(vj + a.eps);
        float Aj = rj * (affine ? gam[j] : 1.0f);
        float Bj = (affine ? bet[j] : 0.0f) - mj * Aj;
        Aj *= sc;
        if (film) {
          const float f1 = fs[j] + 1.0f;
          Aj *= f1;
          Bj = Bj * f1 + fh[j];
        }
        A[j] = Aj;
        Bc[j] = Bj;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = x[j] * A[j] + Bc[j];
      if (a.mode == JEN1_PRO_GN_SILU) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
    }
  } else {
    const float2 rs = *reinterpret_cast<const float2*>(a.ln_rowstats + 2 * grow);
    mean = rs.x * a.inv_count;
    float var = rs.y * a.inv_count - mean * mean;
    var = var < 0.f ? 0.f : var;
    rstd = PRECISE ? 1.0f / sqrtf(var + a.eps) : rsqrtf(var + a.eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = (x[j] - mean) * rstd;
      if (affine) x[j] = x[j] * gam[j] + bet[j];
    }
  }
  store8(reinterpret_cast<T*>(a.y) + (size_t)((unsigned)grow * (unsigned)a.ld_y + (unsigned)c), x);
}

}  // namespace

extern "C" int jen1_norm_apply(const jen1_norm_args* a, void* stream) {
  JEN1_CHECK(a && a->x0 && a->y, "norm_apply: null pointer");
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16, "norm_apply: bad dtype");
  JEN1_CHECK(a->c0 > 0 && a->c0 % 8 == 0 && a->c1 >= 0 && a->c1 % 8 == 0 && (a->c1 == 0 || a->x1), "norm_apply: bad channels");
  JEN1_CHECK(a->ld_y >= a->c0 + a->c1 && a->ld_y % 8 == 0, "norm_apply: bad ld_y");
  const bool gn = (a->mode == JEN1_PRO_GN || a->mode == JEN1_PRO_GN_SILU);
  JEN1_CHECK(gn || a->mode == JEN1_PRO_LN, "norm_apply: bad mode %d", a->mode);
  JEN1_CHECK((a->gamma == nullptr) == (a->beta == nullptr), "norm_apply: gamma and beta come together");
  const int ctot = a->c0 + a->c1;
  NormDev d;
  memset(&d, 0, sizeof(d));
  if (gn) {
    JEN1_CHECK(a->gn_stats0 && a->gamma && a->beta && a->groups >= 1 && a->groups <= 32 && a->cpg >= 1 && a->count >= 1, "norm_apply: incomplete GroupNorm");
    JEN1_CHECK(a->c0 % 32 == 0 && a->c1 % 32 == 0, "norm_apply: GroupNorm sources must be multiples of 32 channels");
    JEN1_CHECK(a->c1 == 0 || (a->gn_stats1 && a->c0 % a->cpg == 0), "norm_apply: bad two-source GroupNorm");
    d.cpf0 = a->c0 / JEN1_FINE_GROUPS;
    d.cpf1 = a->c1 ? a->c1 / JEN1_FINE_GROUPS : 1;
    // fine groups per group; the last group also takes the padding channels (cpg = real channels / groups)
    d.nfg0 = (a->cpg + d.cpf0 - 1) / d.cpf0;
    d.nfg1 = a->c1 ? (a->cpg + d.cpf1 - 1) / d.cpf1 : 0;
    if (a->groups == 1) d.nfg0 = JEN1_FINE_GROUPS;
    JEN1_CHECK((a->cpg % 8 != 0) || (a->cpg % d.cpf0 == 0 && (a->c1 == 0 || a->cpg % d.cpf1 == 0)) || a->groups == 1,
               "norm_apply: group size %d is not a whole number of statistics fine groups", a->cpg);
    d.inv_cpf0 = 1.0f / (float)d.cpf0;
    d.inv_cpf1 = 1.0f / (float)d.cpf1;
  } else {
    JEN1_CHECK(a->ln_rowstats && a->c1 == 0 && a->count >= 1, "norm_apply: incomplete LayerNorm");
  }
  d.x0 = a->x0; d.x1 = a->x1; d.y = a->y; d.st0 = a->gn_stats0; d.st1 = a->gn_stats1;
  d.gamma = a->gamma; d.beta = a->beta;
  d.film = (gn && a->film) ? a->film + a->film_off : nullptr;
  d.film_row = a->film_row; d.film_step = a->film_step; d.ln_rowstats = a->ln_rowstats;
  d.mode = a->mode; d.L = a->L; d.c0 = a->c0; d.c1 = a->c1; d.ld0 = a->ld0; d.ld1 = a->ld1; d.ld_y = a->ld_y;
  d.groups = gn ? a->groups : 1; d.cpg = gn ? a->cpg : ctot;
  d.vpr = ctot / 8;
  d.nvec = a->L * d.vpr;
  JEN1_CHECK(d.nvec < (1 << 20) && (int64_t)a->B * a->L * (a->ld_y > a->ld0 ? a->ld_y : a->ld0) < ((int64_t)1 << 31), "norm_apply: tensor too large for this pre-pass");
  d.film_C = a->film_C; d.film_ld = a->film_ld;
  d.inv_vpr = 1.0f / (float)d.vpr;
  d.inv_cpg = 1.0f / (float)d.cpg;
  d.inv_count = 1.0f / (float)a->count;
  d.eps = a->eps;
  d.src1_scale = a->src1_scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((d.nvec + 255) / 256, a->B);
  if (a->dtype == JEN1_F32) hipLaunchKernelGGL(norm_apply_kernel<float>, grid, dim3(256), 0, s, d);
  else hipLaunchKernelGGL(norm_apply_kernel<bf16_t>, grid, dim3(256), 0, s, d);
  JEN1_HIP(hipGetLastError());
  return 0;
}
