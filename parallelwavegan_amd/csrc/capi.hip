// capi.hip -- error reporting and version queries of the C ABI.
#include "common.h"

#include <mutex>
#include <string>
#include <vector>

namespace pwg {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::mutex g_lds_mu;
static std::vector<std::pair<const void*, size_t>> g_lds_set;
bool lds_limit_is_set(const void* kern, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_lds_mu);
  for (auto& e : g_lds_set)
    if (e.first == kern) {
      if (e.second >= bytes) return true;
      e.second = bytes;
      return false;
    }
  g_lds_set.emplace_back(kern, bytes);
  return false;
}

__global__ void zero_fill_kernel(float* p, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) p[i] = 0.f;
}
void zero_fill(float* p, long n, hipStream_t stream) {
  if (n <= 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((int)blocks), dim3(256), 0, stream, p, n);
}

// ---- debugging aid: PWG_POISON_LDS=1 fills the LDS of every CU with NaN bit patterns before each
// MFMA kernel launch, so that a result depending on stale LDS content (0 * unwritten tile element in a
// contraction) fails deterministically instead of once in a while on a cold GPU ---------------------
__global__ void poison_lds_kernel(float* sink) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < (160 * 1024 - 256) / 4; i += blockDim.x) lds[i] = __int_as_float(0x7fc00000);
  __syncthreads();
  if (sink && threadIdx.x == 0 && lds[17] == 1.0f) sink[0] = 1.f;
}
static int g_poison_lds = -1;  // -1: take PWG_POISON_LDS from the environment at the first launch
void maybe_poison_lds(hipStream_t stream) {
  if (g_poison_lds < 0) g_poison_lds = getenv("PWG_POISON_LDS") != nullptr ? 1 : 0;
  if (!g_poison_lds) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr = true;
  }
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024 - 256, stream, (float*)nullptr);
}

// ---- per-launch event timing ------------------------------------------------------------
struct ProfRec {
  const char* kernel;
  double flops, bytes;
  hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_open;  // launches not yet folded into the totals
struct ProfTotal {
  std::string kernel;
  double ms = 0, flops = 0, bytes = 0;
  long launches = 0;
};
static std::vector<ProfTotal> g_prof_totals;

bool prof_enabled() { return g_prof_on; }

const char* prof_shape_name(const char* family, const char* fmt, ...) {
  static const bool shapes = getenv("PWG_PROF_SHAPES") != nullptr;
  if (!shapes || !g_prof_on) return family;
  char buf[256];
  int n = snprintf(buf, sizeof(buf), "%s ", family);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf + n, sizeof(buf) - n, fmt, ap);
  va_end(ap);
  static std::mutex mu;
  static std::vector<std::string*> pool;  // interned: ProfRec keeps the pointer
  std::lock_guard<std::mutex> lk(mu);
  for (auto* s : pool)
    if (*s == buf) return s->c_str();
  pool.push_back(new std::string(buf));
  return pool.back()->c_str();
}

void prof_record(hipStream_t stream, const char* kernel, double flops, double bytes, bool begin) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (begin) {
    ProfRec r{kernel, flops, bytes, nullptr, nullptr};
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, stream);
    g_prof_open.push_back(r);
  } else {
    for (size_t i = g_prof_open.size(); i-- > 0;)
      if (g_prof_open[i].kernel == kernel) {  // innermost open scope of this kernel family
        (void)hipEventRecord(g_prof_open[i].e1, stream);
        break;
      }
  }
}

static void prof_fold() {
  for (auto& r : g_prof_open) {
    float ms = 0.f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    ProfTotal* t = nullptr;
    for (auto& c : g_prof_totals)
      if (c.kernel == r.kernel) t = &c;
    if (!t) {
      g_prof_totals.emplace_back();
      t = &g_prof_totals.back();
      t->kernel = r.kernel;
    }
    t->ms += ms;
    t->flops += r.flops;
    t->bytes += r.bytes;
    t->launches += 1;
  }
  g_prof_open.clear();
}
}  // namespace pwg

extern "C" int pwg_debug_poison_lds(int on) {
  const int was = pwg::g_poison_lds > 0 ? 1 : 0;
  pwg::g_poison_lds = on ? 1 : 0;
  return was;
}

extern "C" int pwg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(pwg::g_prof_mu);
  pwg::g_prof_on = on != 0;
  return PWG_OK;
}
extern "C" int pwg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(pwg::g_prof_mu);
  pwg::prof_fold();
  pwg::g_prof_totals.clear();
  return PWG_OK;
}
extern "C" int pwg_prof_num_kernels(void) {
  std::lock_guard<std::mutex> lk(pwg::g_prof_mu);
  pwg::prof_fold();
  return (int)pwg::g_prof_totals.size();
}
extern "C" int pwg_prof_get(int32_t idx, char* name, size_t name_cap, double* total_ms, int64_t* launches,
                            double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(pwg::g_prof_mu);
  pwg::prof_fold();
  PWG_REQUIRE(idx >= 0 && idx < (int)pwg::g_prof_totals.size(), PWG_ERR_BAD_SHAPE, "prof_get: index %d out of range", idx);
  const auto& t = pwg::g_prof_totals[idx];
  if (name && name_cap) snprintf(name, name_cap, "%s", t.kernel.c_str());
  if (total_ms) *total_ms = t.ms;
  if (launches) *launches = t.launches;
  if (flops) *flops = t.flops;
  if (bytes) *bytes = t.bytes;
  return PWG_OK;
}

extern "C" const char* pwg_last_error(void) { return pwg::g_err; }
extern "C" int pwg_abi_version(void) { return 11; }
extern "C" int pwg_target_arch(void) { return 950; }
