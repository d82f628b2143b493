// gemm_lt.hip -- the plain fp32 GEMMs of the training step (Linear forward / input gradient / weight gradient of
// DistSageConv, python/layers/dist_sageconv.py:34,80) as direct hipBLASLt calls behind the C ABI (cslicer_aggr.h,
// csl_gemm_f32): what a host without a tensor framework calls.  These are library GEMMs, not kernels of this project;
// what this file adds is the host side:
//   * one cached plan per shape (matmul descriptor, layouts, algorithm): a call is one hipblasLtMatmul, ~5 us of host
//     time (a framework's dispatcher + tuning lookup: 25-35 us per GEMM, eight GEMMs per step);
//   * the algorithm of a shape is chosen by TIMING the library's candidates on its first use (the default heuristic
//     is up to 2x off for the tall-skinny shapes of a minibatch: profiles/gemm_shape_probe.py);
//   * row-major operands, bias and ReLU in the GEMM epilogue.
// The Python trainer keeps torch's GEMMs with recorded TunableOp selections by default (they search every solution and
// come out 6 % faster in sum; CSLICER_DIRECT_GEMMS=1 switches: cslicer/splitgnn.py).
// The library is bound at run time (dlopen of the libhipblaslt.so.1 the process already has, else ROCm's): linking it
// would pull a second copy next to the one a host framework ships.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

struct LtApi {
  void* lib = nullptr;
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatrixLayoutSetAttribute) LayoutSet = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
  decltype(&hipblasLtGetVersion) GetVersion = nullptr;
  // hipblaslt_ext (C++ entry points, bound by their mangled names; optional: exhaustive tuning and recorded plans)
  hipblasStatus_t (*AllAlgos)(hipblasLtHandle_t, int /*GemmType*/, hipblasOperation_t, hipblasOperation_t, hipDataType,
                              hipDataType, hipDataType, hipDataType, hipblasComputeType_t,
                              std::vector<hipblasLtMatmulHeuristicResult_t>&) = nullptr;
  hipblasStatus_t (*AlgosFromIndex)(hipblasLtHandle_t, std::vector<int>&,
                                    std::vector<hipblasLtMatmulHeuristicResult_t>&) = nullptr;
  hipblasStatus_t (*IsSupported)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, const void*, hipblasLtMatrixLayout_t,
                                 hipblasLtMatrixLayout_t, const void*, hipblasLtMatrixLayout_t, hipblasLtMatrixLayout_t,
                                 hipblasLtMatmulAlgo_t&, size_t&) = nullptr;
  int (*IndexFromAlgo)(hipblasLtMatmulAlgo_t&) = nullptr;
  bool ext() const { return AllAlgos && AlgosFromIndex && IsSupported && IndexFromAlgo; }
};

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;  // the library's A, B (column-major) and C = D
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  float us = 0.f;  // the chosen algorithm's time when the shape was tuned
  int n_tried = 0;
};

// transa, transb, m, n, k, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, epilogue
typedef std::tuple<int, int, long long, long long, long long, long long, long long, long long, int, long long, long long,
                   long long, int>
    Key;

struct State {
  std::mutex mu;
  LtApi api;
  bool api_tried = false;
  hipblasLtHandle_t handle = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int device = -1;            // the device the handle and the workspace were created on
  void* stream = nullptr;     // the ONE stream GEMMs are issued on (the workspace is shared by every plan)
  std::map<Key, Plan> plans;
  // the algorithm last chosen for a shape CLASS (the shape with its long dimension -- the rows of a minibatch layer:
  // m, or k of a weight gradient -- blanked): a new row count of a known class takes it without timing anything
  std::map<Key, std::pair<hipblasLtMatmulAlgo_t, size_t>> classes;
  // recorded plans (csl_gemm_load_plans): shape class -> the library's solution index, valid for one library version
  std::map<Key, int> recorded;
  int recorded_version = -1;
  char err[256] = {0};
};
State g;

constexpr size_t WS_BYTES = 128ull << 20;
constexpr int MAX_ALGOS = 48;

template <typename F>
bool bind(void* lib, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(lib, name));
  return f != nullptr;
}

bool load_api() {
  if (g.api_tried) return g.api.lib != nullptr;
  g.api_tried = true;
  const char* names[] = {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) {
    snprintf(g.err, sizeof(g.err), "hipBLASLt not found: %s", dlerror());
    return false;
  }
  LtApi& a = g.api;
  const bool ok = bind(lib, "hipblasLtCreate", a.Create) && bind(lib, "hipblasLtMatrixLayoutCreate", a.LayoutCreate) &&
                  bind(lib, "hipblasLtMatrixLayoutSetAttribute", a.LayoutSet) &&
                  bind(lib, "hipblasLtMatmulDescCreate", a.DescCreate) &&
                  bind(lib, "hipblasLtMatmulDescDestroy", a.DescDestroy) &&
                  bind(lib, "hipblasLtMatrixLayoutDestroy", a.LayoutDestroy) &&
                  bind(lib, "hipblasLtMatmulDescSetAttribute", a.DescSet) &&
                  bind(lib, "hipblasLtMatmulPreferenceCreate", a.PrefCreate) &&
                  bind(lib, "hipblasLtMatmulPreferenceSetAttribute", a.PrefSet) &&
                  bind(lib, "hipblasLtMatmulPreferenceDestroy", a.PrefDestroy) &&
                  bind(lib, "hipblasLtMatmulAlgoGetHeuristic", a.Heuristic) && bind(lib, "hipblasLtMatmul", a.Matmul);
  if (!ok) {
    snprintf(g.err, sizeof(g.err), "hipBLASLt lacks an entry point this file needs");
    return false;
  }
  bind(lib, "hipblasLtGetVersion", a.GetVersion);
  bind(lib, "_ZN13hipblaslt_ext11getAllAlgosEPvNS_8GemmTypeE18hipblasOperation_tS2_11hipDataTypeS3_S3_S3_20hipblasComputeType_tRSt6vectorI33_hipblasLtMatmulHeuristicResult_tSaIS6_EE", a.AllAlgos);
  bind(lib, "_ZN13hipblaslt_ext17getAlgosFromIndexEPvRSt6vectorIiSaIiEERS1_I33_hipblasLtMatmulHeuristicResult_tSaIS5_EE", a.AlgosFromIndex);
  bind(lib, "_ZN13hipblaslt_ext21matmulIsAlgoSupportedEPvP27hipblasLtMatmulDescOpaque_tPKvP29hipblasLtMatrixLayoutOpaque_tS6_S4_S6_S6_R22_hipblasLtMatmulAlgo_tRm", a.IsSupported);
  bind(lib, "_ZN13hipblaslt_ext16getIndexFromAlgoER22_hipblasLtMatmulAlgo_t", a.IndexFromAlgo);
  a.lib = lib;
  return true;
}

int lib_version() {
  int v = -1;
  if (g.api.GetVersion && g.handle) g.api.GetVersion(g.handle, &v);
  return v;
}

#define LT(x)                                                                                 \
  do {                                                                                        \
    hipblasStatus_t s_ = (x);                                                                 \
    if (s_ != HIPBLAS_STATUS_SUCCESS) {                                                       \
      snprintf(g.err, sizeof(g.err), "%s -> hipblas status %d (%s:%d)", #x, (int)s_, __FILE__, __LINE__); \
      return CSL_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)
#define HIPOK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      snprintf(g.err, sizeof(g.err), "%s -> %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return CSL_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)

int run(const Plan& p, const float* A, const float* B, float* C, const float* bias, hipStream_t st) {
  const float alpha = 1.f, beta = 0.f;
  if (bias) LT(g.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  // row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T: the library's first operand is OUR B
  LT(g.api.Matmul(g.handle, p.desc, &alpha, B, p.la, A, p.lb, &beta, C, p.lc, C, p.lc, &p.algo, g.ws, p.ws, st));
  return 0;
}

// *ran: the GEMM itself has been issued (with the algorithm the plan keeps) as part of making the plan
int make_plan_impl(const Key& key, Plan& p, const float* A, const float* B, float* C, const float* bias, hipStream_t st,
                   bool* ran);
int make_plan(const Key& key, Plan& p, const float* A, const float* B, float* C, const float* bias, hipStream_t st,
              bool* ran) {
  const int r = make_plan_impl(key, p, A, B, C, bias, st, ran);
  if (r) {  // a plan that could not be made leaves nothing behind
    if (p.la) g.api.LayoutDestroy(p.la);
    if (p.lb) g.api.LayoutDestroy(p.lb);
    if (p.lc) g.api.LayoutDestroy(p.lc);
    if (p.desc) g.api.DescDestroy(p.desc);
    p = Plan();
  }
  return r;
}
int make_plan_impl(const Key& key, Plan& p, const float* A, const float* B, float* C, const float* bias, hipStream_t st,
                   bool* ran) {
  *ran = false;
  int transa, transb, batch, epi;
  long long m, n, k, lda, ldb, ldc, sa, sb, sc;
  std::tie(transa, transb, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, epi) = key;
  LtApi& a = g.api;
  LT(a.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  // our B, row-major [k, n] (or [n, k] when transb) = a column-major n x k (k x n) matrix: the library's A operand
  const int32_t op_first = transb ? HIPBLAS_OP_T : HIPBLAS_OP_N, op_second = transa ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LT(a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_first, sizeof(op_first)));
  LT(a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_second, sizeof(op_second)));
  const uint32_t e = (uint32_t)epi;
  LT(a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &e, sizeof(e)));
  if (bias) LT(a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  LT(a.LayoutCreate(&p.la, HIP_R_32F, transb ? (uint64_t)k : (uint64_t)n, transb ? (uint64_t)n : (uint64_t)k, ldb));
  LT(a.LayoutCreate(&p.lb, HIP_R_32F, transa ? (uint64_t)m : (uint64_t)k, transa ? (uint64_t)k : (uint64_t)m, lda));
  LT(a.LayoutCreate(&p.lc, HIP_R_32F, (uint64_t)n, (uint64_t)m, ldc));
  if (batch > 1) {
    const int32_t bc = batch;
    const int64_t s_first = sb, s_second = sa, s_c = sc;
    LT(a.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
    LT(a.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
    LT(a.LayoutSet(p.lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
    LT(a.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &s_first, sizeof(s_first)));
    LT(a.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &s_second, sizeof(s_second)));
    LT(a.LayoutSet(p.lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &s_c, sizeof(s_c)));
  }
  // The long dimension (the rows: m, or k of a weight gradient) is blanked to a SIZE BUCKET, not to nothing: 0 = 4096 rows
  // and more (what the recorded plans were tuned on; for a weight gradient: slabs of 1024 rows and more, the 1.9 k-row slabs of
  // a batch-1024 step included), 2 = 1024..4095 rows, 1 = fewer.  With one class for every row count the 1 k-row layers
  // -- the top layer of every model, every layer of a 128-seed rank or replica step -- ran the 128 x 192 tile kernel chosen
  // for 8 k rows on eight workgroups (28 us for 0.27 GFLOP: profiles/r3_rank/rank128_kernels.md).
  const long long small = transa ? (k < 1024 ? 1 : 0) : (m < 1024 ? 1 : (m < 4096 ? 2 : 0));
  const Key cls = transa ? Key(transa, transb, m, n, small, lda, ldb, ldc, batch, 0, 0, sc, epi)
                         : Key(transa, transb, small, n, k, lda, ldb, ldc, batch, 0, sb, 0, epi);
  {
    auto ci = g.classes.find(cls);
    if (ci != g.classes.end()) {
      p.algo = ci->second.first;
      p.ws = ci->second.second;
      if (run(p, A, B, C, bias, st) == 0) {  // (else: the library refuses it for this size -- time afresh)
        *ran = true;
        return 0;
      }
    }
  }
  const float one = 1.f, zero = 0.f;
  if (a.ext() && g.recorded_version == lib_version()) {
    auto ri = g.recorded.find(cls);
    if (ri != g.recorded.end()) {
      std::vector<int> idx(1, ri->second);
      std::vector<hipblasLtMatmulHeuristicResult_t> rr;
      size_t wsz = 0;
      if (a.AlgosFromIndex(g.handle, idx, rr) == HIPBLAS_STATUS_SUCCESS && !rr.empty() &&
          a.IsSupported(g.handle, p.desc, &one, p.la, p.lb, &zero, p.lc, p.lc, rr[0].algo, wsz) == HIPBLAS_STATUS_SUCCESS &&
          wsz <= g.ws_bytes) {
        p.algo = rr[0].algo;
        p.ws = wsz;
        if (run(p, A, B, C, bias, st) == 0) {
          g.classes[cls] = std::make_pair(p.algo, p.ws);
          *ran = true;
          return 0;
        }
      }
    }
  }
  const char* tune_env = getenv("CSLICER_GEMM_TUNE");
  const bool tune = !(tune_env && tune_env[0] == '0');
  const bool exhaustive = tune_env && tune_env[0] == 'a' && a.ext();  // "all": every solution of the library
  hipblasLtMatmulPreference_t pref = nullptr;
  LT(a.PrefCreate(&pref));
  const uint64_t wsb = g.ws_bytes;
  LT(a.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)));
  std::vector<hipblasLtMatmulHeuristicResult_t> res(MAX_ALGOS);
  int got = 0;
  const hipblasStatus_t hs = a.Heuristic(g.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, MAX_ALGOS, res.data(), &got);
  a.PrefDestroy(pref);
  if (hs != HIPBLAS_STATUS_SUCCESS || got < 1) {
    snprintf(g.err, sizeof(g.err), "hipBLASLt has no algorithm for this GEMM (status %d, %d candidates)", (int)hs, got);
    return CSL_E_HIP;
  }
  res.resize(got);
  if (exhaustive) {
    // what a framework's offline tuner does: every solution the library has for these types, filtered by support
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    if (a.AllAlgos(g.handle, 1 /* HIPBLASLT_GEMM */, (hipblasOperation_t)op_first, (hipblasOperation_t)op_second, HIP_R_32F,
                   HIP_R_32F, HIP_R_32F, HIP_R_32F, HIPBLAS_COMPUTE_32F, all) == HIPBLAS_STATUS_SUCCESS) {
      for (auto& r : all) {
        size_t wsz = 0;
        if (a.IsSupported(g.handle, p.desc, &one, p.la, p.lb, &zero, p.lc, p.lc, r.algo, wsz) != HIPBLAS_STATUS_SUCCESS) continue;
        r.workspaceSize = wsz;
        r.state = HIPBLAS_STATUS_SUCCESS;
        res.push_back(r);
      }
    }
    got = (int)res.size();
  }
  int best = -1;
  float best_us = 0.f;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct EvGuard {   // the tuning loop has early error returns: the events go with the scope
    hipEvent_t &a, &b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      a = b = nullptr;
    }
  } ev_guard{e0, e1};
  if (tune) {
    HIPOK(hipEventCreate(&e0));
    HIPOK(hipEventCreate(&e1));
  }
  for (int i = 0; i < got; i++) {
    if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > g.ws_bytes) continue;
    if (!tune) {
      best = i;
      break;
    }
    p.algo = res[i].algo;
    p.ws = res[i].workspaceSize;
    HIPOK(hipEventRecord(e0, st));
    if (run(p, A, B, C, bias, st) != 0) continue;  // (a candidate the library then refuses is skipped)
    HIPOK(hipEventRecord(e1, st));
    if (hipStreamSynchronize(st) != hipSuccess) return CSL_E_HIP;
    if (best >= 0) {  // a first run three times slower than the best so far is not timed further
      float ms1 = 0.f;
      HIPOK(hipEventElapsedTime(&ms1, e0, e1));
      if (ms1 * 1e3f > 3.f * best_us + 20.f) continue;
    }
    const int reps = 3;
    HIPOK(hipEventRecord(e0, st));
    bool ok = true;
    for (int r = 0; r < reps && ok; r++) ok = run(p, A, B, C, bias, st) == 0;
    HIPOK(hipEventRecord(e1, st));
    HIPOK(hipEventSynchronize(e1));
    if (!ok) continue;
    float ms = 0.f;
    HIPOK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / reps;
    p.n_tried++;
    if (best < 0 || us < best_us) {
      best = i;
      best_us = us;
    }
  }
  if (best < 0) {
    snprintf(g.err, sizeof(g.err), "none of hipBLASLt's %d candidates ran for this GEMM", got);
    return CSL_E_HIP;
  }
  p.algo = res[best].algo;
  p.ws = res[best].workspaceSize;
  p.us = best_us;
  g.classes[cls] = std::make_pair(p.algo, p.ws);
  if (getenv("CSLICER_GEMM_LOG"))
    fprintf(stderr, "[csl_gemm] %c%c m=%lld n=%lld k=%lld batch=%d epilogue=%d: candidate %d of %d (%d timed), %.1f us\n",
            transa ? 'T' : 'N', transb ? 'T' : 'N', m, n, k, batch, epi, best, got, p.n_tried, best_us);
  return 0;
}

// out[c] = sum_b slabs[b][c]
__global__ __launch_bounds__(256) void k_sum_slabs(const float4* __restrict__ slabs, long long n4, int nslab,
                                                   float4* __restrict__ out) {
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
  if (c >= n4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int b = 0;
  for (; b + 4 <= nslab; b += 4) {  // four loads in flight
    const float4 v0 = slabs[(long long)b * n4 + c], v1 = slabs[(long long)(b + 1) * n4 + c];
    const float4 v2 = slabs[(long long)(b + 2) * n4 + c], v3 = slabs[(long long)(b + 3) * n4 + c];
    acc.x += (v0.x + v1.x) + (v2.x + v3.x), acc.y += (v0.y + v1.y) + (v2.y + v3.y);
    acc.z += (v0.z + v1.z) + (v2.z + v3.z), acc.w += (v0.w + v1.w) + (v2.w + v3.w);
  }
  for (; b < nslab; b++) {
    const float4 v = slabs[(long long)b * n4 + c];
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  out[c] = acc;
}

}  // namespace

extern "C" {

const char* csl_gemm_last_error(void) { return g.err; }

int csl_gemm_f32(int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, const float* A, int64_t lda,
                 int64_t stride_a, const float* B, int64_t ldb, int64_t stride_b, float* C, int64_t ldc, int64_t stride_c,
                 int32_t batch, const float* bias, int32_t relu, void* stream) {
  if (m < 0 || n < 0 || k < 0 || batch < 1) return CSL_E_INVALID;
  if (m == 0 || n == 0) return 0;
  if (!C) return CSL_E_INVALID;
  if (k == 0) {  // an empty reduction (a rank whose share of the minibatch is empty): zeros (+ bias / ReLU not needed)
    if (bias || relu) return CSL_E_INVALID;
    for (int32_t b = 0; b < batch; b++)
      if (hipMemset2DAsync(C + (size_t)b * stride_c, sizeof(float) * ldc, 0, sizeof(float) * n, m, (hipStream_t)stream) != hipSuccess)
        return CSL_E_HIP;
    return 0;
  }
  if (!A || !B) return CSL_E_INVALID;
  if (lda < (transa ? m : k) || ldb < (transb ? k : n) || ldc < n) return CSL_E_INVALID;
  std::lock_guard<std::mutex> lock(g.mu);
  if (!load_api()) return CSL_E_HIP;
  {
    // one handle and one workspace per process: every GEMM must come from the device they were created on and from
    // ONE stream (two GEMMs on different streams would share the workspace; a second device would get the first one's
    // memory).  The trainer's contract (one GPU per process, one training stream); anything else is refused, loudly.
    int dev = -1;
    HIPOK(hipGetDevice(&dev));
    if (!g.handle) {
      LT(g.api.Create(&g.handle));
      HIPOK(hipMalloc(&g.ws, WS_BYTES));
      g.ws_bytes = WS_BYTES;
      g.device = dev;
      g.stream = stream;
    } else if (dev != g.device || stream != g.stream) {
      snprintf(g.err, sizeof(g.err), "csl_gemm_f32 is bound to device %d and the stream of its first call (one workspace): "
               "called from device %d / another stream", g.device, dev);
      return CSL_E_STATE;
    }
  }
  const int epi = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                       : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
  const Key key(transa ? 1 : 0, transb ? 1 : 0, m, n, k, lda, ldb, ldc, batch, batch > 1 ? stride_a : 0,
                batch > 1 ? stride_b : 0, batch > 1 ? stride_c : 0, epi);
  auto it = g.plans.find(key);
  if (it == g.plans.end()) {
    Plan p;
    bool ran = false;
    const int r = make_plan(key, p, A, B, C, bias, (hipStream_t)stream, &ran);
    if (r) return r;
    it = g.plans.emplace(key, p).first;
    if (ran) return 0;
  }
  return run(it->second, A, B, C, bias, (hipStream_t)stream);
}

/* Recorded plans: one line per shape class, "ta tb m n k lda ldb ldc batch sa sb sc epilogue index" (the long dimension
 * is its size bucket -- 0: 4096 rows and more (weight-gradient slabs: 1024 and more), 2: 1024..4095, 1: fewer --, the strides that follow it are 0), after a header naming the library version the solution indices belong to. */
int csl_gemm_save_plans(const char* path) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (!path || !g.handle || !g.api.ext()) return CSL_E_STATE;
  FILE* f = fopen(path, "w");
  if (!f) return CSL_E_INVALID;
  fprintf(f, "# csl_gemm_f32 plans: hipblaslt %d\n", lib_version());
  for (auto& kv : g.classes) {
    int ta, tb, batch, epi;
    long long m, n, k, lda, ldb, ldc, sa, sb, sc;
    std::tie(ta, tb, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, epi) = kv.first;
    hipblasLtMatmulAlgo_t algo = kv.second.first;
    fprintf(f, "%d %d %lld %lld %lld %lld %lld %lld %d %lld %lld %lld %d %d\n", ta, tb, m, n, k, lda, ldb, ldc, batch, sa, sb,
            sc, epi, g.api.IndexFromAlgo(algo));
  }
  fclose(f);
  return 0;
}

int csl_gemm_load_plans(const char* path) {
  std::lock_guard<std::mutex> lock(g.mu);
  if (!path) return CSL_E_INVALID;
  FILE* f = fopen(path, "r");
  if (!f) return CSL_E_INVALID;
  char line[512];
  int version = -1, n_read = 0;
  std::map<Key, int> rec;
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#') {
      sscanf(line, "# csl_gemm_f32 plans: hipblaslt %d", &version);
      continue;
    }
    int ta, tb, batch, epi, idx;
    long long m, n, k, lda, ldb, ldc, sa, sb, sc;
    if (sscanf(line, "%d %d %lld %lld %lld %lld %lld %lld %d %lld %lld %lld %d %d", &ta, &tb, &m, &n, &k, &lda, &ldb, &ldc,
               &batch, &sa, &sb, &sc, &epi, &idx) == 14 && idx >= 0) {
      rec[Key(ta, tb, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, epi)] = idx;
      n_read++;
    }
  }
  fclose(f);
  g.recorded.swap(rec);
  g.recorded_version = version;   // compared with the loaded library's version when a plan is made
  return n_read;
}

int csl_sum_slabs_f32(const float* slabs, int64_t n, int32_t n_slabs, float* out, void* stream) {
  if (n < 0 || n_slabs < 1 || n % 4 != 0) return CSL_E_INVALID;
  if (n == 0) return 0;
  if (!slabs || !out || ((uintptr_t)slabs & 15) || ((uintptr_t)out & 15)) return CSL_E_INVALID;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(slabs), n4, (int)n_slabs, reinterpret_cast<float4*>(out));
  return hipGetLastError() == hipSuccess ? 0 : CSL_E_HIP;
}

}  // extern "C"
