// Optimiser step of the JEN-1 trainer on gfx950:
//   nn.utils.clip_grad_norm_(model.parameters(), 0.7) -> torch.optim.AdamW.step()
//   (/root/reference/trainer.py:144-149, /root/reference/train.py:56-60: lr 3e-5, betas (0.9, 0.95), weight_decay 0.1,
//    one parameter group: the decay applies to every parameter).
// Elementwise over 296.5 M float32 parameters + gradient + two moments: 28 bytes per parameter, purely HBM-bound.
//   jen1_grad_sqnorm : sum of squares of the flat gradient -> one device float (block partials, summed in a fixed order by
//                      the last block to arrive: bit-reproducible, so data-parallel replicas clip identically)
//   jen1_adamw_step  : reads that device scalar, so clip + AdamW is ONE pass with no host synchronisation:
//                      g' = g * min(1, max_norm / (||g|| + 1e-6));  p *= 1 - lr wd;  m, v updates;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
//   A non-finite gradient norm skips the step (GradScaler semantics of trainer.py:147 when fp16 is on).
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));   // streamed once: keep it out of the caches
  return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int OPT_THREADS = 256;
constexpr int OPT_VEC_PER_THREAD = 4;        // float4 vectors per thread per grid-stride pass

// Deterministic: block partials go to a fixed slot each, the last block to arrive sums the slots in a fixed order.
// (With one float atomic per block the sum depended on the arrival order, the clip coefficient differed in its last
// bits from run to run -- and between the ranks of a data-parallel job, whose replicas then drift apart.)
constexpr int OPT_MAX_BLOCKS = 4096;
__device__ float g_sq_partials[OPT_MAX_BLOCKS];
__device__ unsigned g_sq_ticket;

// `partials` (OPT_MAX_BLOCKS floats) and `ticket` are the caller's scratch (jen1_grad_sqnorm_ws: one per optimiser, so that two
// optimisers / streams never share them) or the library's own (jen1_grad_sqnorm: one call at a time per device)
__global__ __launch_bounds__(OPT_THREADS) void grad_sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out,
                                                                   float* __restrict__ partials, unsigned* __restrict__ ticket_p) {
  __shared__ float red[OPT_THREADS / 64];
  __shared__ int last_s;
  const int64_t nv = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * OPT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < nv; i += stride * OPT_VEC_PER_THREAD) {
    float4 v[OPT_VEC_PER_THREAD];
#pragma unroll
    for (int u = 0; u < OPT_VEC_PER_THREAD; ++u) {
      const int64_t j = i + u * stride;
      v[u] = j < nv ? nt_load4(g4 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < OPT_VEC_PER_THREAD; ++u) s += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float t = g[(nv << 2) + threadIdx.x];
    s += t * t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
    __hip_atomic_store(&partials[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(ticket_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (ticket == gridDim.x - 1) ? 1 : 0;
    if (last_s) __hip_atomic_store(ticket_p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_s) return;
  // the last block: fixed-order sum of the gridDim.x partials (thread t takes slots t, t + 256, ...; then a fixed tree)
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += OPT_THREADS)
    t += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < OPT_THREADS / 64; ++w) tot += red[w];
    out[0] += tot;
  }
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* gnorm_sq;     // device scalar (sum of squares over ALL parameters) or NULL: no clipping
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, max_norm, bc1, inv_sqrt_bc2;
  int32_t skip_nonfinite;
  const int32_t* step_counter;      // device-side count of the steps TAKEN so far (nullptr: bc1 / inv_sqrt_bc2 come from the host)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
  g *= coef;
  p *= 1.0f - a.lr * a.weight_decay;
  m = a.beta1 * m + (1.0f - a.beta1) * g;                      // torch: exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(AdamArgs a) {
  float coef = 1.0f;
  if (a.step_counter) {
    // bias corrections of step (taken so far + 1), in double like the host form; a skipped step (skip_nonfinite) does not count,
    // which is what GradScaler.step does with torch's AdamW (trainer.py:146)
    const double t = (double)(a.step_counter[0] + 1);
    a.bc1 = (float)(1.0 - pow((double)a.beta1, t));
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, t)));
  }
  if (a.gnorm_sq) {
    const float nrm = sqrtf(a.gnorm_sq[0]);
    if (a.skip_nonfinite && !(nrm <= 3.0e38f)) return;          // inf / nan gradients: leave parameters and moments untouched
    if (a.max_norm > 0.f) {
      const float c = a.max_norm / (nrm + 1e-6f);
      coef = c < 1.0f ? c : 1.0f;
    }
  }
  const int64_t nv = a.n >> 2;
  float4* p4 = reinterpret_cast<float4*>(a.p);
  const float4* g4 = reinterpret_cast<const float4*>(a.g);
  float4* m4 = reinterpret_cast<float4*>(a.m);
  float4* v4 = reinterpret_cast<float4*>(a.v);
  const int64_t stride = (int64_t)gridDim.x * OPT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < nv; i += stride) {
    float4 p = p4[i], m = m4[i], v = v4[i];
    const float4 g = nt_load4(g4 + i);
    adam1(p.x, g.x, m.x, v.x, a, coef);
    adam1(p.y, g.y, m.y, v.y, a, coef);
    adam1(p.z, g.z, m.z, v.z, a, coef);
    adam1(p.w, g.w, m.w, v.w, a, coef);
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int64_t j = (nv << 2) + threadIdx.x;
    float p = a.p[j], m = a.m[j], v = a.v[j];
    adam1(p, a.g[j], m, v, a, coef);
    a.p[j] = p; a.m[j] = m; a.v[j] = v;
  }
}

// behind adamw_kernel on the same stream: count the step unless it was skipped
__global__ void adamw_advance_kernel(int32_t* step_counter, const float* gnorm_sq, int skip_nonfinite) {
  if (skip_nonfinite && gnorm_sq && !(sqrtf(gnorm_sq[0]) <= 3.0e38f)) return;
  step_counter[0] += 1;
}

}  // namespace

static int launch_sqnorm(const float* g, int64_t n, float* out, float* partials, unsigned* ticket, void* stream) {
  JEN1_CHECK(g && out && n > 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "grad_sqnorm: null / unaligned pointer or empty tensor");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t nv = n >> 2;
  int64_t blocks = (nv + OPT_THREADS * OPT_VEC_PER_THREAD - 1) / (OPT_THREADS * OPT_VEC_PER_THREAD);
  blocks = blocks < 1 ? 1 : (blocks > OPT_MAX_BLOCKS ? OPT_MAX_BLOCKS : blocks);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, g, n, out, partials, ticket);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_grad_sqnorm(const float* g, int64_t n, float* out, void* stream) {
  float* partials = nullptr;
  unsigned* ticket = nullptr;
  JEN1_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&partials), HIP_SYMBOL(g_sq_partials)));
  JEN1_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ticket), HIP_SYMBOL(g_sq_ticket)));
  return launch_sqnorm(g, n, out, partials, ticket, stream);
}

extern "C" int64_t jen1_grad_sqnorm_scratch_bytes(void) { return (int64_t)(OPT_MAX_BLOCKS + 64) * 4; }

extern "C" int jen1_grad_sqnorm_ws(const float* g, int64_t n, float* out, void* scratch, void* stream) {
  JEN1_CHECK(scratch && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "grad_sqnorm_ws: null / unaligned scratch");
  float* partials = reinterpret_cast<float*>(scratch);
  return launch_sqnorm(g, n, out, partials, reinterpret_cast<unsigned*>(partials + OPT_MAX_BLOCKS), stream);
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, int32_t* step_counter, const float* gnorm_sq, float max_norm, int skip_nonfinite,
                        void* stream);

extern "C" int jen1_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, const float* gnorm_sq, float max_norm, int skip_nonfinite, void* stream) {
  JEN1_CHECK(step >= 1, "adamw_step: bad step");
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, gnorm_sq, max_norm, skip_nonfinite, stream);
}

extern "C" int jen1_adamw_step_counted(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                       float eps, float weight_decay, int32_t* step_counter, const float* gnorm_sq, float max_norm,
                                       int skip_nonfinite, void* stream) {
  JEN1_CHECK(step_counter, "adamw_step_counted: null step counter");
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, 1, step_counter, gnorm_sq, max_norm, skip_nonfinite, stream);
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, int32_t* step_counter, const float* gnorm_sq, float max_norm, int skip_nonfinite,
                        void* stream) {
  JEN1_CHECK(p && g && m && v && n > 0, "adamw_step: null pointer or empty tensor");
  JEN1_CHECK(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
             "adamw_step: buffers must be 16-byte aligned");
  JEN1_CHECK(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adamw_step: bad betas");
  AdamArgs a;
  a.step_counter = step_counter;
  a.p = p; a.g = g; a.m = m; a.v = v; a.gnorm_sq = gnorm_sq; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.max_norm = max_norm;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  a.skip_nonfinite = skip_nonfinite;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t nv = n >> 2;
  int64_t blocks = (nv + OPT_THREADS - 1) / OPT_THREADS;
  blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, a);
  if (step_counter) hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, s, step_counter, gnorm_sq, skip_nonfinite);
  JEN1_HIP(hipGetLastError());
  return 0;
}
