// libperiod_hip.so -- C ABI (include/periodhip.h) over the gfx950 kernels in ph_kernels.h.
// Host side only: argument checks, device staging for host-pointer calls, small integer
// tables, launch geometry.  No compute happens on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/periodhip.h"
#include "ph_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define PH_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(e_ == hipErrorOutOfMemory ? PH_E_NOMEM : PH_E_HIP, "%s failed: %s", #call, \
                  hipGetErrorString(e_));                                                    \
  } while (0)

#define PH_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != PH_OK) return rc_; \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct TableSlot {
  DevBuf dev;
  std::vector<int32_t> host;  // last uploaded content (skip identical re-uploads)
  bool valid = false;
};

enum { T_PLIST, T_ORTH_OFF, T_ORTH_Q, T_FAC_OFF, T_FAC_Q, T_AUX0, T_AUX1, T_AUX2, T_AUX3, T_COUNT };
enum { B_IN, B_OUT0, B_OUT1, B_OUT2, B_OUT3, B_OUT4, B_WS0, B_WS1, B_GEN0, B_GBUF, B_GWIN, B_COUNT };

}  // namespace

struct ph_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 0;
  int lds_limit = 0;
  DevBuf buf[B_COUNT];
  TableSlot tab[T_COUNT];
  // per-period fold geometry for the tuned sweeps, cached for the last (N, max_p)
  DevBuf geom;
  DevBuf geomf;  // the same table in float (ph::PGeomF), for the window-pair screen
  DevBuf kapf;   // kappa_q of k_small_to_large_pair's flag test (ph::s2l_kappa, rounded up to float)
  int geom_n = -1, geom_max_p = -1;
  bool step1_pair = true;  // PH_STEP1_PAIR=0: always the one-window fp64 kernel for m_best step 1
  bool s2l_pair = true;    // PH_S2L_PAIR=0: always the one-window kernel for small_to_large
  bool bc_pair = true;     // PH_BC_PAIR=0: always the one-window kernel for best_correlation
  bool pair_chain = true;  // PH_PAIR_CHAIN=0: the window-pair kernels take the periods below 64 one pass each
  DevBuf twid;  // cos/sin(2 pi k / L), k < L, of the last best_frequency win_size
  int twid_len = -1;
  DevBuf bs_tab;  // Bluestein tables of the last (win_size, min(N, win_size)): M twiddles, chirp, FFT of the wrapped chirp
  int bs_L = -1, bs_M0 = -1;
  DevBuf plan;  // pass plan of the norm sweeps, cached for the last (p_lo, p_hi)
  int plan_lo = -1, plan_hi = -1, plan_n = 0, plan_m = -1;
  int plan_max_m = 4;  // largest row-class count a pass may use (PH_PLAN_MAX_M overrides: 1, 2 or 4)
  int sweep_block = ph::kBlockWide;  // threads per workgroup of the sweep kernels (PH_SWEEP_BLOCK overrides)
  int step1_block = 0;               // k_mbest_step1 only: 0 = automatic (PH_STEP1_BLOCK overrides, <= 1024)
  int qo_block = 1024;               // k_qo_find threads per workgroup (PH_QO_BLOCK overrides)
  int s2l_block = 1024;              // k_small_to_large_pair threads per workgroup (PH_S2L_BLOCK overrides, >= 512)
  bool qo_hbm_window = false;        // PH_QO_HBM_WINDOW=1: keep the residual of k_qo_find in HBM even when it fits LDS
  // optional per-kernel HIP-event timing (ph_profile_*)
  bool prof_on = false;
  int prof_n = 0;
  std::vector<hipEvent_t> prof_ev;  // 2 events per recorded launch
  std::vector<const char*> prof_name;
};

namespace {

int ensure(ph_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return PH_OK;
  if (b.p) {
    PH_HIP(hipStreamSynchronize(c->stream));
    PH_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = std::max<size_t>(bytes, 256);
  PH_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  return PH_OK;
}

// Upload a small int table (host pointer) into a cached device slot.
int upload_table(ph_ctx* c, int slot, const int32_t* src, size_t n, const int** dev_out) {
  TableSlot& t = c->tab[slot];
  if (n == 0) {
    PH_TRY(ensure(c, t.dev, 16));
    *dev_out = static_cast<const int*>(t.dev.p);
    return PH_OK;
  }
  if (t.valid && t.host.size() == n && std::memcmp(t.host.data(), src, n * sizeof(int32_t)) == 0) {
    *dev_out = static_cast<const int*>(t.dev.p);
    return PH_OK;
  }
  t.valid = false;
  // the previous content may still be in use by queued kernels
  PH_HIP(hipStreamSynchronize(c->stream));
  PH_TRY(ensure(c, t.dev, n * sizeof(int32_t) + 256));  // 64 readable words behind the table (wave-wide list reads)
  t.host.assign(src, src + n);
  PH_HIP(hipMemcpyAsync(t.dev.p, t.host.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  PH_HIP(hipStreamSynchronize(c->stream));
  t.valid = true;
  *dev_out = static_cast<const int*>(t.dev.p);
  return PH_OK;
}

size_t elem_size(int dtype) { return dtype == PH_F64 ? 8 : 4; }

// Pass plan of a norm sweep over [p_lo, p_hi]: every period is produced exactly once, either
// by its own pass or as 2p / 4p of a smaller base period (see PassPlan in ph_device.h).
// Periods below 64 use the row-split path one at a time (m = 0).
// `chains` (window-pair kernels): the periods up to 64 are taken in chains L, L/2, L/4, ... -- one row-split pass at
// L yields them all (m = 8 + number of periods, see pair_chain_small in ph_pair.h) -- instead of one pass each.
std::vector<ph::PassPlan> build_plan(int p_lo, int p_hi, int max_m, bool mixed = true, bool chains = false) {
  std::vector<ph::PassPlan> host;
  std::vector<char> covered((size_t)p_hi + 1, 0);
  if (chains) {
    for (int L = std::min(p_hi, 64); L >= p_lo; --L) {
      if (covered[L]) continue;
      int n = 0;
      for (int q = L; q >= p_lo && q >= 1 && !covered[q]; q >>= 1) {
        covered[q] = 1;
        n += 1;
        if (q & 1) break;
      }
      host.push_back(ph::PassPlan{L, 8 + n});
    }
    std::reverse(host.begin(), host.end());
  }
  for (int p = p_lo; p <= p_hi; ++p) {
    if (p < 64 && !chains) {
      host.push_back(ph::PassPlan{p, 0});
      continue;
    }
    if (covered[p]) continue;
    int m = (4LL * p <= p_hi) ? 4 : (2LL * p <= p_hi) ? 2 : 1;
    m = std::min(m, max_m);
    for (int d = 1; d <= m; d *= 2) covered[(size_t)d * p] = 1;
    host.push_back(ph::PassPlan{p, m});
  }
  // Interleave the pass types evenly (entry i of a type with n entries gets the key (i + 0.5) / n):
  // the waves of a CU walk the plan in step, and a mix of LDS-heavy single passes and VALU-heavy
  // multi-class passes overlaps better than a phase of each.
  if (mixed && !std::getenv("PH_PLAN_SORTED")) {
    int count[5] = {0, 0, 0, 0, 0}, seen[5] = {0, 0, 0, 0, 0};
    for (const auto& e : host) count[e.m >= 8 ? 0 : e.m] += 1;
    std::vector<std::pair<double, size_t>> key(host.size());
    for (size_t i = 0; i < host.size(); ++i) {
      const int m = host[i].m >= 8 ? 0 : host[i].m;
      key[i] = {(seen[m] + 0.5) / count[m], i};
      seen[m] += 1;
    }
    std::stable_sort(key.begin(), key.end());
    std::vector<ph::PassPlan> mixed(host.size());
    for (size_t i = 0; i < host.size(); ++i) mixed[i] = host[key[i].second];
    host.swap(mixed);
  }
  if (const char* only = std::getenv("PH_PLAN_ONLY_M")) {  // profiling aid: keep one pass type (results incomplete)
    const int want = std::atoi(only);
    std::vector<ph::PassPlan> kept;
    for (const auto& e : host)
      if (e.m == want) kept.push_back(e);
    host.swap(kept);
  }
  return host;
}

// `mixed`: interleave the pass types (kernels whose workgroups split the plan in chunks: a mix of LDS-heavy
// and VALU-heavy passes overlaps better); otherwise ascending base period, i.e. the expensive multi-class
// passes first and the cheap few-row singles last -- what a workgroup that walks the WHOLE plan between two
// barriers wants (k_mbest_step1: shorter tail before the argmax barrier, -3 %).
int prepare_plan(ph_ctx* c, int p_lo, int p_hi, const ph::PassPlan** out, int* n_pass, int max_m = 4, bool mixed = true,
                 bool chains = false) {
  max_m = std::min(max_m, c->plan_max_m);
  chains = chains && c->pair_chain;
  if (!mixed) max_m += 8;  // cache key
  if (chains) max_m += 16;
  if (c->plan.p && c->plan_lo == p_lo && c->plan_hi == p_hi && c->plan_m == max_m) {
    *out = static_cast<const ph::PassPlan*>(c->plan.p);
    *n_pass = c->plan_n;
    return PH_OK;
  }
  const std::vector<ph::PassPlan> host = build_plan(p_lo, p_hi, max_m & 7, mixed, chains);
  PH_HIP(hipStreamSynchronize(c->stream));
  PH_TRY(ensure(c, c->plan, std::max<size_t>(1, host.size()) * sizeof(ph::PassPlan)));
  if (!host.empty())
    PH_HIP(hipMemcpyAsync(c->plan.p, host.data(), host.size() * sizeof(ph::PassPlan), hipMemcpyHostToDevice,
                          c->stream));
  PH_HIP(hipStreamSynchronize(c->stream));
  c->plan_lo = p_lo;
  c->plan_hi = p_hi;
  c->plan_m = max_m;
  c->plan_n = (int)host.size();
  *out = static_cast<const ph::PassPlan*>(c->plan.p);
  *n_pass = c->plan_n;
  return PH_OK;
}

// Dense table geom[p], p in [0, max_p]: rows, nfull and the reciprocal counts of period p.
int prepare_geom(ph_ctx* c, int N, int max_p, const ph::PGeom** out) {
  if (c->geom.p && c->geom_n == N && c->geom_max_p >= max_p) {
    *out = static_cast<const ph::PGeom*>(c->geom.p);
    return PH_OK;
  }
  std::vector<ph::PGeom> host((size_t)max_p + 1);
  host[0] = ph::PGeom{0, 0, 0.0, 0.0};
  for (int p = 1; p <= max_p; ++p) {
    const int rows = (N + p - 1) / p;
    const int shortn = rows * p - N;
    host[p] = ph::PGeom{rows, p - shortn, 1.0 / rows, rows > 1 ? 1.0 / (rows - 1) : 0.0};
  }
  PH_HIP(hipStreamSynchronize(c->stream));
  PH_TRY(ensure(c, c->geom, host.size() * sizeof(ph::PGeom)));
  PH_HIP(hipMemcpyAsync(c->geom.p, host.data(), host.size() * sizeof(ph::PGeom), hipMemcpyHostToDevice, c->stream));
  std::vector<ph::PGeomF> hostf(host.size());
  for (size_t p = 0; p < host.size(); ++p)
    hostf[p] = ph::PGeomF{host[p].rows, host[p].nfull, (float)host[p].w_full, (float)host[p].w_short};
  PH_TRY(ensure(c, c->geomf, hostf.size() * sizeof(ph::PGeomF)));
  PH_HIP(hipMemcpyAsync(c->geomf.p, hostf.data(), hostf.size() * sizeof(ph::PGeomF), hipMemcpyHostToDevice, c->stream));
  std::vector<float> hostk(host.size(), 0.0f);
  for (size_t p = 1; p < host.size(); ++p) {
    const double k = ph::s2l_kappa(N, host[p].rows, (int)p);
    float f = (float)k;
    if ((double)f < k) f = std::nextafterf(f, INFINITY);
    hostk[p] = f;
  }
  PH_TRY(ensure(c, c->kapf, hostk.size() * sizeof(float)));
  PH_HIP(hipMemcpyAsync(c->kapf.p, hostk.data(), hostk.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PH_HIP(hipStreamSynchronize(c->stream));
  c->geom_n = N;
  c->geom_max_p = max_p;
  *out = static_cast<const ph::PGeom*>(c->geom.p);
  return PH_OK;
}

int check_common(ph_ctx* c, const void* x, int dtype, int64_t W, int N) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  if (!x) return fail(PH_E_ARG, "x is NULL");
  if (dtype != PH_F64 && dtype != PH_F32) return fail(PH_E_ARG, "dtype must be PH_F64 or PH_F32");
  if (W < 1 || W > 0x7fffffffLL / 64) return fail(PH_E_ARG, "W=%lld out of range", (long long)W);
  if (N < 1) return fail(PH_E_ARG, "N=%d must be >= 1", N);
  return PH_OK;
}

int check_lds(ph_ctx* c, size_t bytes, int N, const char* what) {
  if (bytes > (size_t)c->lds_limit)
    return fail(PH_E_ARG, "%s: window of N=%d needs %zu B of LDS, device limit is %d B", what, N, bytes,
                c->lds_limit);
  return PH_OK;
}

// Second window-sized buffer (materialised projections): in LDS when both fit, otherwise in an HBM
// workspace of `blocks` x N elements.  `lds` comes in with the second buffer included.
int place_second_buffer(ph_ctx* c, size_t* lds, bool needed, size_t buf_bytes, int64_t blocks, void** gbuf) {
  *gbuf = nullptr;
  if (!needed || *lds <= (size_t)c->lds_limit) return PH_OK;
  *lds -= ((buf_bytes + 15) & ~size_t(15));
  PH_TRY(ensure(c, c->buf[B_GBUF], (size_t)blocks * buf_bytes));
  *gbuf = c->buf[B_GBUF].p;
  return PH_OK;
}

// Window buffer: LDS when it fits (after the second buffer has been placed), otherwise an HBM
// workspace of `blocks` slices that the kernels' <T, false> instantiations fold through L2.
// `lds` comes in with the window buffer included.
int place_window(ph_ctx* c, size_t* lds, size_t win_len, size_t sz, int64_t blocks, void** gwin) {
  *gwin = nullptr;
  if (*lds <= (size_t)c->lds_limit) return PH_OK;
  *lds -= ph::carve_bytes(win_len, sz);
  PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)blocks * ph::win_stride(win_len) * sz));
  *gwin = c->buf[B_GWIN].p;
  return PH_OK;
}

// f(T{}, std::bool_constant<window in LDS>{}) for the runtime element type / window placement.
template <typename F>
int dispatch(int dtype, bool lds_window, F&& f) {
  if (dtype == PH_F64) return lds_window ? f(double{}, std::true_type{}) : f(double{}, std::false_type{});
  return lds_window ? f(float{}, std::true_type{}) : f(float{}, std::false_type{});
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(PH_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e));
  }
  return PH_OK;
}

constexpr int kProfCap = 256;

// Brackets one kernel launch with HIP events on the context's stream when profiling is on.
struct ProfScope {
  ph_ctx* c;
  bool live;
  ProfScope(ph_ctx* ctx, const char* name) : c(ctx), live(ctx->prof_on && ctx->prof_n < kProfCap) {
    if (!live) return;
    if ((int)c->prof_ev.size() < 2 * (c->prof_n + 1)) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        live = false;
        return;
      }
      c->prof_ev.push_back(a);
      c->prof_ev.push_back(b);
      c->prof_name.push_back(name);
    }
    c->prof_name[c->prof_n] = name;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream);
  }
  ~ProfScope() {
    if (!live) return;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream);
    c->prof_n += 1;
  }
};

int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PH_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return PH_OK;
}

// Orth tables (only when PH_FLAG_ORTH) and validation of their content.
int prepare_orth(ph_ctx* c, unsigned flags, const int32_t* off, const int32_t* q, int table_max_p, int need_p,
                 ph::Tables* tb) {
  tb->orth_off = tb->orth_q = nullptr;
  if (!(flags & PH_FLAG_ORTH)) return PH_OK;
  if (!off || !q) return fail(PH_E_ARG, "PH_FLAG_ORTH needs orth_off/orth_q tables");
  if (table_max_p < need_p) return fail(PH_E_ARG, "orth tables cover p <= %d, need %d", table_max_p, need_p);
  const size_t n_off = (size_t)table_max_p + 2;
  if (off[0] != 0) return fail(PH_E_ARG, "orth_off[0] must be 0");
  for (size_t i = 0; i + 1 < n_off; ++i)
    if (off[i + 1] < off[i]) return fail(PH_E_ARG, "orth_off not monotone at %zu", i);
  const size_t n_q = (size_t)off[n_off - 1];
  for (size_t i = 0; i < n_q; ++i)
    if (q[i] < 1) return fail(PH_E_ARG, "orth_q[%zu]=%d must be >= 1", i, q[i]);
  PH_TRY(upload_table(c, T_ORTH_OFF, off, n_off, &tb->orth_off));
  PH_TRY(upload_table(c, T_ORTH_Q, q, n_q, &tb->orth_q));
  return PH_OK;
}

int prepare_fac(ph_ctx* c, const int32_t* off, const int32_t* q, int table_max_p, int need_p, ph::Tables* tb) {
  if (!off || !q) return fail(PH_E_ARG, "fac_off/fac_q tables are required");
  if (table_max_p < need_p) return fail(PH_E_ARG, "factor tables cover p <= %d, need %d", table_max_p, need_p);
  const size_t n_off = (size_t)table_max_p + 2;
  if (off[0] != 0) return fail(PH_E_ARG, "fac_off[0] must be 0");
  for (size_t i = 0; i + 1 < n_off; ++i)
    if (off[i + 1] < off[i]) return fail(PH_E_ARG, "fac_off not monotone at %zu", i);
  const size_t n_q = (size_t)off[n_off - 1];
  for (size_t i = 0; i < n_q; ++i)
    if (q[i] < 1 || q[i] > table_max_p) return fail(PH_E_ARG, "fac_q[%zu]=%d out of range", i, q[i]);
  PH_TRY(upload_table(c, T_FAC_OFF, off, n_off, &tb->fac_off));
  PH_TRY(upload_table(c, T_FAC_Q, q, n_q, &tb->fac_q));
  return PH_OK;
}

// Host-pointer staging: `Stage` maps each user array to the pointer the kernel uses.
struct Stage {
  ph_ctx* c;
  bool device;
  struct Out {
    void* user;
    void* dev;
    size_t bytes;
  };
  std::vector<Out> outs;
  Stage(ph_ctx* ctx, unsigned flags) : c(ctx), device(flags & PH_FLAG_DEVICE) {}
  int in(const void* user, size_t bytes, const void** dev) {
    if (device) {
      *dev = user;
      return PH_OK;
    }
    PH_TRY(ensure(c, c->buf[B_IN], bytes));
    PH_HIP(hipMemcpyAsync(c->buf[B_IN].p, user, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = c->buf[B_IN].p;
    return PH_OK;
  }
  int out(int slot, void* user, size_t bytes, void** dev) {
    if (!user) {
      *dev = nullptr;
      return PH_OK;
    }
    if (device) {
      *dev = user;
      return PH_OK;
    }
    PH_TRY(ensure(c, c->buf[slot], bytes));
    *dev = c->buf[slot].p;
    outs.push_back({user, *dev, bytes});
    return PH_OK;
  }
  int finish() {
    if (device) return PH_OK;
    for (const Out& o : outs)
      PH_HIP(hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream));
    PH_HIP(hipStreamSynchronize(c->stream));
    return PH_OK;
  }
};

int pick_chunks(ph_ctx* c, int64_t W, int items, int min_items_per_chunk) {
  const int64_t target = (int64_t)c->num_cu * 8;  // >= 8 workgroups per CU in flight/queued
  int64_t chunks = (target + W - 1) / W;
  const int64_t max_chunks = std::max(1, items / std::max(1, min_items_per_chunk));
  chunks = std::max<int64_t>(1, std::min(chunks, max_chunks));
  return (int)chunks;
}

using ph::carve_bytes;
using ph::kBlock;
using ph::kBlockWide;
using ph::kPad;
using ph::kMaxWaves;
using ph::kRedDoubles;

// LDS layout of k_qo_find: bookkeeping, the pair table, the weights, and the six work vectors + sample counts of the
// conjugate-gradient solve (one slot per dictionary row and one per block).  The residual window stays in LDS when
// it fits; the work vectors then OVERLAY it if it is large enough (the window is dead during a solve) -- 79 KB per
// workgroup for N = 16384 fp32 with kcap = 1024, two workgroups per CU -- and sit behind it otherwise.  Windows
// that do not fit (long fp64 windows) move to the HBM workspace and the sweeps read them through L2.
int qo_lds_layout(ph_ctx* c, size_t sz, int N, int max_length, int kcap, size_t* lds_out, bool* lds_window_out,
                  bool* overlay_out) {
  const size_t kv = (size_t)kcap + ph::kQoMaxBlocks;
  const size_t fixed = carve_bytes(kRedDoubles, 8) + carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4) +
                       2 * carve_bytes(ph::kQoMaxBlocks, 4) + carve_bytes(ph::kQoMaxBlocks + 1, 4) +
                       carve_bytes(ph::kQoMaxBlocks, 8) + carve_bytes((max_length + 32) / 32, 4) +
                       carve_bytes(2 * ph::kQoPairTab * ph::kQoPairTab, 4) + carve_bytes(ph::kQoMaxBlocks, 4) +
                       carve_bytes(ph::kQoMaxBlocks + 1, 4) + carve_bytes(6 * kMaxWaves, 8) + carve_bytes(kv, 8);
  const size_t solver = 6 * carve_bytes(kv, 8) + carve_bytes(kv, 4);
  const size_t win = carve_bytes(N + kPad, sz);
  const size_t limit = (size_t)c->lds_limit;
  if (fixed + solver > limit)
    return fail(PH_E_ARG, "ph_qo_find_periods: kcap=%d needs %zu B of LDS for the solver (limit %d B)", kcap,
                fixed + solver, c->lds_limit);
  const bool overlay = win >= solver;
  const size_t with_window = fixed + (overlay ? win : win + solver);
  const bool lds_window = !c->qo_hbm_window && with_window <= limit;
  *lds_out = lds_window ? with_window : fixed + solver;
  *lds_window_out = lds_window;
  *overlay_out = lds_window && overlay;
  return PH_OK;
}

// LDS of k_mbest_step1_pair: the pair window, one fp64 staging buffer, bookkeeping of two windows.
size_t pair_lds_bytes(int N, int num, int P) {
  return 2 * carve_bytes(N + kPad, 8) + carve_bytes(kRedDoubles, 8) + carve_bytes(kMaxWaves, 8) +
         carve_bytes(kMaxWaves, 4) + carve_bytes(2 * num, 8) + carve_bytes(2 * num, 4) +
         carve_bytes(2 * ((P + 31) / 32), 4) + carve_bytes(2 * ph::kPairListCap, 4) + carve_bytes(16, 4) +
         carve_bytes(4, 8) + carve_bytes(ph::kPairSmallP, 8) + carve_bytes(ph::kPairSplitW, 8);
}

// The window-pair screen serves fp64 windows, plain projection, candidate periods below N, when the pair window
// and the staging buffer fit the LDS; everything else runs k_mbest_step1.
bool pair_eligible(const ph_ctx* c, int dtype, int N, int num, int min_length, int max_length, unsigned flags) {
  const bool general = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  return c->step1_pair && dtype == PH_F64 && !general && max_length < N && min_length <= max_length &&
         pair_lds_bytes(N, num, max_length - min_length + 1) <= (size_t)c->lds_limit;
}

}  // namespace

// =========================================================================================
extern "C" {

int ph_version(void) { return PH_VERSION; }

#ifdef PH_CLOCKS
// diagnostic builds only (not part of the C ABI): workgroup stamps of the last k_small_to_large_pair launch
int ph_debug_stamps(long long* dst, int n_wg) {
  PH_HIP(hipDeviceSynchronize());
  PH_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(ph::g_ph_stamps), sizeof(long long) * 4 * (size_t)n_wg));
  return PH_OK;
}
#endif

const char* ph_last_error(void) { return g_err.c_str(); }

int ph_device_count(int* count) {
  if (!count) return fail(PH_E_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(PH_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return PH_OK;
}

int ph_create(int device, ph_ctx** out) {
  if (!out) return fail(PH_E_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  PH_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(PH_E_ARG, "device %d not in [0, %d)", device, n);
  PH_HIP(hipSetDevice(device));
  ph_ctx* c = new (std::nothrow) ph_ctx();
  if (!c) return fail(PH_E_NOMEM, "out of host memory");
  c->device = device;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    delete c;
    return fail(PH_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  c->num_cu = prop.multiProcessorCount;
  // gfx950 has 160 KiB of LDS per CU and one workgroup may use all of it; take the largest
  // figure the runtime reports (an oversize launch fails cleanly with a launch error).
  size_t lds = prop.sharedMemPerBlock;
  lds = std::max(lds, prop.sharedMemPerBlockOptin);
  lds = std::max(lds, prop.maxSharedMemoryPerMultiProcessor);
  c->lds_limit = (int)std::min<size_t>(lds, 160 * 1024);
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
    delete c;
    return fail(PH_E_HIP, "stream/event creation: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  if (const char* e = std::getenv("PH_PLAN_MAX_M")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4) c->plan_max_m = v;
  }
  if (const char* e = std::getenv("PH_STEP1_PAIR")) c->step1_pair = std::atoi(e) != 0;
  if (const char* e = std::getenv("PH_S2L_PAIR")) c->s2l_pair = std::atoi(e) != 0;
  if (const char* e = std::getenv("PH_BC_PAIR")) c->bc_pair = std::atoi(e) != 0;
  if (const char* e = std::getenv("PH_PAIR_CHAIN")) c->pair_chain = std::atoi(e) != 0;
  if (const char* e = std::getenv("PH_STEP1_BLOCK")) {
    const int v = std::atoi(e);
    if (v >= 64 && v <= 1024 && v % 64 == 0) c->step1_block = v;
  }
  if (const char* e = std::getenv("PH_S2L_BLOCK")) {
    const int v = std::atoi(e);
    if (v >= 512 && v <= 1024 && v % 64 == 0) c->s2l_block = v;
  }
  if (const char* e = std::getenv("PH_QO_BLOCK")) {
    const int v = std::atoi(e);
    if (v >= 64 && v <= 1024 && v % 64 == 0) c->qo_block = v;
  }
  if (std::getenv("PH_QO_HBM_WINDOW")) c->qo_hbm_window = true;
  if (const char* e = std::getenv("PH_SWEEP_BLOCK")) {
    const int v = std::atoi(e);
    if (v >= 64 && v <= 512 && v % 64 == 0) c->sweep_block = v;
  }
  *out = c;
  return PH_OK;
}

int ph_destroy(ph_ctx* c) {
  if (!c) return PH_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (DevBuf& b : c->buf)
    if (b.p) (void)hipFree(b.p);
  for (TableSlot& t : c->tab)
    if (t.dev.p) (void)hipFree(t.dev.p);
  if (c->geom.p) (void)hipFree(c->geom.p);
  if (c->geomf.p) (void)hipFree(c->geomf.p);
  if (c->kapf.p) (void)hipFree(c->kapf.p);
  if (c->plan.p) (void)hipFree(c->plan.p);
  if (c->twid.p) (void)hipFree(c->twid.p);
  if (c->bs_tab.p) (void)hipFree(c->bs_tab.p);
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return PH_OK;
}

int ph_set_stream(ph_ctx* c, void* hip_stream) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  // drain the old stream, but rebind even if that fails (a borrowed stream may have been destroyed:
  // the context must not stay stuck on it); the error is reported after the switch
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (hip_stream == PH_STREAM_DEFAULT)
    c->stream = nullptr;  // the legacy default stream
  else
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(PH_E_HIP, "ph_set_stream: draining the previous stream failed (%s); the new stream is bound",
                hipGetErrorString(e));
  }
  return PH_OK;
}

int ph_sweep_plan_info(ph_ctx* c, int p_lo, int p_hi, int* n_pass, int* n_periods) {
  if (!c || !n_pass || !n_periods) return fail(PH_E_ARG, "NULL argument");
  if (p_lo < 1 || p_hi < p_lo) return fail(PH_E_ARG, "need 1 <= p_lo <= p_hi (got %d, %d)", p_lo, p_hi);
  *n_pass = (int)build_plan(p_lo, p_hi, c->plan_max_m, true, c->pair_chain).size();  // the plan ph_sweep's norm modes run
  *n_periods = p_hi - p_lo + 1;
  return PH_OK;
}

int ph_m_best_info(ph_ctx* c, int dtype, int N, int num, int min_length, int max_length, unsigned flags,
                   int* windows_per_workgroup, int* lds_bytes_per_sample) {
  if (!c || !windows_per_workgroup || !lds_bytes_per_sample) return fail(PH_E_ARG, "NULL argument");
  if (dtype != PH_F64 && dtype != PH_F32) return fail(PH_E_ARG, "dtype must be PH_F64 or PH_F32");
  if (max_length < 0) max_length = N / 3;
  const bool pair = pair_eligible(c, dtype, N, num, min_length, max_length, flags);
  *windows_per_workgroup = pair ? 2 : 1;
  *lds_bytes_per_sample = pair ? 8 : (int)elem_size(dtype);
  return PH_OK;
}

int ph_m_best_plan_info(ph_ctx* c, int dtype, int N, int num, int min_length, int max_length, unsigned flags, int* n_pass,
                        int* n_periods) {
  if (!c || !n_pass || !n_periods) return fail(PH_E_ARG, "NULL argument");
  if (dtype != PH_F64 && dtype != PH_F32) return fail(PH_E_ARG, "dtype must be PH_F64 or PH_F32");
  if (max_length < 0) max_length = N / 3;
  if (min_length < 1 || max_length < min_length)
    return fail(PH_E_ARG, "need 1 <= min_length <= max_length (got %d, %d)", min_length, max_length);
  const bool pair = pair_eligible(c, dtype, N, num, min_length, max_length, flags);
  *n_pass = (int)build_plan(min_length, max_length, c->plan_max_m, false, pair && c->pair_chain).size();
  *n_periods = max_length - min_length + 1;
  return PH_OK;
}

int ph_sync(ph_ctx* c) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  PH_HIP(hipStreamSynchronize(c->stream));
  return PH_OK;
}

int ph_timer_begin(ph_ctx* c) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  PH_HIP(hipEventRecord(c->ev0, c->stream));
  return PH_OK;
}

int ph_timer_end(ph_ctx* c, float* ms) {
  if (!c || !ms) return fail(PH_E_ARG, "NULL argument");
  PH_HIP(hipEventRecord(c->ev1, c->stream));
  PH_HIP(hipEventSynchronize(c->ev1));
  PH_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return PH_OK;
}

int ph_profile_enable(ph_ctx* c, int on) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  c->prof_on = on != 0;
  c->prof_n = 0;
  return PH_OK;
}

int ph_profile_read(ph_ctx* c, float* ms, int cap, int* count) {
  if (!c || !count) return fail(PH_E_ARG, "NULL argument");
  PH_HIP(hipStreamSynchronize(c->stream));
  *count = c->prof_n;
  for (int i = 0; i < c->prof_n && i < cap && ms; ++i)
    PH_HIP(hipEventElapsedTime(&ms[i], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
  return PH_OK;
}

const char* ph_profile_name(ph_ctx* c, int i) {
  if (!c || i < 0 || i >= c->prof_n) return "";
  return c->prof_name[i];
}

int ph_device_info(ph_ctx* c, int* num_cu, int* lds_bytes) {
  if (!c) return fail(PH_E_ARG, "ctx is NULL");
  if (num_cu) *num_cu = c->num_cu;
  if (lds_bytes) *lds_bytes = c->lds_limit;
  return PH_OK;
}

int ph_max_window(ph_ctx* c, int dtype, unsigned flags, int* max_n) {
  if (!c || !max_n) return fail(PH_E_ARG, "NULL argument");
  const size_t sz = elem_size(dtype);
  const size_t overhead = 8192;  // reduction scratch, bookkeeping arrays
  (void)flags;  // a second (projection) buffer moves to an HBM workspace when LDS cannot hold two
  *max_n = (int)(((size_t)c->lds_limit - overhead) / sz);
  return PH_OK;
}

// ----------------------------------------------------------------------------- K1
int ph_project_batch(ph_ctx* c, const void* x, int dtype, int64_t W, int N, const int32_t* p_list, int n_p,
                     const int32_t* orth_off, const int32_t* orth_q, int table_max_p, unsigned flags,
                     void* out) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!p_list || n_p < 1 || !out) return fail(PH_E_ARG, "p_list/out NULL or n_p < 1");
  int pmax = 1;
  for (int k = 0; k < n_p; ++k) {
    if (p_list[k] < 1) return fail(PH_E_ARG, "p_list[%d]=%d must be >= 1", k, p_list[k]);
    pmax = std::max(pmax, p_list[k]);
  }
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const int scratch_len = (flags & PH_FLAG_ORTH) ? N : std::min(pmax, N);
  size_t lds = carve_bytes(N, sz) + carve_bytes(scratch_len, sz);
  const int chunks = pick_chunks(c, W, n_p, 1);
  void* gbuf;
  PH_TRY(place_second_buffer(c, &lds, true, (size_t)scratch_len * sz, W * chunks, &gbuf));
  void* gwin;
  PH_TRY(place_window(c, &lds, N, sz, W * chunks, &gwin));
  PH_TRY(check_lds(c, lds, N, "ph_project_batch"));
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, flags, orth_off, orth_q, table_max_p, pmax, &tb));
  const int* d_plist;
  PH_TRY(upload_table(c, T_PLIST, p_list, n_p, &d_plist));
  Stage st(c, flags);
  const void* dx;
  void* dout;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * n_p * N * sz, &dout));
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH | PH_FLAG_SINGLE);
  if (flags & PH_FLAG_SINGLE) PH_HIP(hipMemsetAsync(dout, 0, (size_t)W * n_p * N * sz, c->stream));
  const dim3 grid((unsigned)(W * chunks));
  PH_TRY(dispatch(dtype, !gwin, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_project_batch<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_project_batch");
    hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, c->stream, (const T*)dx, N, d_plist, n_p, chunks, kflags, tb,
                       scratch_len, (T*)gbuf, (T*)gwin, (T*)dout);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_project_batch"));
  return st.finish();
}

// ----------------------------------------------------------------------------- K2
int ph_sweep(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int p_lo, int p_hi, int mode,
             const int32_t* orth_off, const int32_t* orth_q, int table_max_p, unsigned flags, double* out) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!out) return fail(PH_E_ARG, "out is NULL");
  if (p_lo < 1 || p_hi < p_lo) return fail(PH_E_ARG, "need 1 <= p_lo <= p_hi (got %d, %d)", p_lo, p_hi);
  if (mode < 0 || mode > 2) return fail(PH_E_ARG, "mode %d unknown", mode);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const bool general = (flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH)) && mode != PH_SWEEP_MAXABS;
  size_t lds = carve_bytes(N + kPad, sz) + (general ? carve_bytes(N, sz) : 0) + carve_bytes(kRedDoubles, 8) + carve_bytes(4, 4);
  const int P = p_hi - p_lo + 1;
  // long windows (at most two workgroups per CU): 16 wavefronts per workgroup
  const int sweep_block = (3 * lds > (size_t)c->lds_limit && c->sweep_block == ph::kBlockWide) ? 1024 : c->sweep_block;
  const int chunks = pick_chunks(c, W, P, 8 * (sweep_block / 64));
  void* gbuf;
  PH_TRY(place_second_buffer(c, &lds, general, (size_t)N * sz, W * chunks, &gbuf));
  void* gwin;
  PH_TRY(place_window(c, &lds, N + kPad, sz, W * chunks, &gwin));
  PH_TRY(check_lds(c, lds, N, "ph_sweep"));
  const ph::PGeom* geom;
  PH_TRY(prepare_geom(c, N, p_hi, &geom));
  const ph::PassPlan* plan;
  int n_pass;
  // norm modes: the periods up to 64 in chains (one row-split pass yields L, L/2, ...: wave_chain_small)
  PH_TRY(prepare_plan(c, p_lo, p_hi, &plan, &n_pass, 4, true, true));
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, general ? flags : 0u, orth_off, orth_q, table_max_p, p_hi, &tb));
  Stage st(c, flags);
  const void* dx;
  void* dout;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * P * sizeof(double), &dout));
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const dim3 grid((unsigned)(W * chunks));
  PH_TRY(dispatch(dtype, !gwin, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_sweep<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_sweep");
    hipLaunchKernelGGL(kernel, grid, dim3(sweep_block), lds, c->stream, (const T*)dx, N, p_lo, p_hi, mode, chunks,
                       kflags, tb, geom, plan, n_pass, (T*)gbuf, (T*)gwin, (double*)dout);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_sweep"));
  return st.finish();
}

// ----------------------------------------------------------------------------- m_best
int ph_m_best(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int num, int min_length, int max_length,
              int gamma, const int32_t* orth_off, const int32_t* orth_q, const int32_t* fac_off,
              const int32_t* fac_q, int table_max_p, unsigned flags, uint32_t* periods, double* powers,
              void* bases, int32_t* status, int32_t* n_sweeps) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!periods || !powers || !bases || !status) return fail(PH_E_ARG, "output pointer is NULL");
  if (num < 1 || num > 4096) return fail(PH_E_ARG, "num=%d must be in [1, 4096]", num);
  if (max_length < 0) max_length = N / 3;  // Periods.py:485-486
  if (min_length < 1 || max_length < min_length)
    return fail(PH_E_ARG, "need 1 <= min_length <= max_length (got %d, %d)", min_length, max_length);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const bool general = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const int P = max_length - min_length + 1;
  size_t lds1 = carve_bytes(N + kPad, sz) + (general ? carve_bytes(N, sz) : 0) + carve_bytes(kRedDoubles, 8) +
                carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4) + carve_bytes(num, 8) +
                carve_bytes(num, 4) + carve_bytes((P + 31) / 32, 4);
  if (!fac_off || !fac_q) return fail(PH_E_ARG, "fac_off/fac_q tables are required");
  if (table_max_p < max_length)
    return fail(PH_E_ARG, "factor tables cover p <= %d, need %d", table_max_p, max_length);
  int max_fac = 1;  // most proper divisors any candidate period has
  for (int q = 0; q <= max_length; ++q) max_fac = std::max(max_fac, fac_off[q + 1] - fac_off[q]);
  size_t lds2 = carve_bytes(N + kPad, sz) + carve_bytes(N, sz) + carve_bytes(kRedDoubles, 8) +
                carve_bytes(num, 8) + carve_bytes(num, 4) + carve_bytes(max_fac, 8) + carve_bytes(max_fac, 4) +
                carve_bytes(kMaxWaves, sizeof(ph::PGeom)) + 3 * carve_bytes(num, 4) + carve_bytes(std::max(max_fac, 64), 4);
  // LDS for the means of a short winning period (split_row_means): only when the window stays in LDS beside it
  const size_t lds_small = carve_bytes(ph::kPairSmallP, sz) + carve_bytes(ph::kPairSplitW, sz);
  const int small_means = (!general && lds1 + lds_small <= (size_t)c->lds_limit) ? 1 : 0;
  if (small_means) lds1 += lds_small;
  void *gbuf1, *gbuf2;
  PH_TRY(place_second_buffer(c, &lds1, general, (size_t)N * sz, W, &gbuf1));
  // step 2 materialises a projection only when a row is split (rare) or in the trunc/orth modes:
  // in plain mode that buffer always lives in the HBM workspace, which lets four workgroups of
  // eight wavefronts share a CU instead of two of four
  if (general) {
    PH_TRY(place_second_buffer(c, &lds2, true, (size_t)N * sz, W, &gbuf2));
  } else {
    lds2 -= carve_bytes(N, sz);
    PH_TRY(ensure(c, c->buf[B_GBUF], (size_t)W * N * sz));
    gbuf2 = c->buf[B_GBUF].p;
  }
  void *gwin1, *gwin2;  // the two kernels run back to back on one stream and may share the workspace
  PH_TRY(place_window(c, &lds1, N + kPad, sz, W, &gwin1));
  PH_TRY(place_window(c, &lds2, N + kPad, sz, W, &gwin2));
  PH_TRY(check_lds(c, std::max(lds1, lds2), N, "ph_m_best"));
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, flags, orth_off, orth_q, table_max_p, max_length, &tb));
  PH_TRY(prepare_fac(c, fac_off, fac_q, table_max_p, max_length, &tb));
  const ph::PGeom* geom;
  PH_TRY(prepare_geom(c, N, max_length, &geom));
  const ph::PassPlan* plan;
  int n_pass;
  // Window-pair screen (k_mbest_step1_pair): fp64 windows, plain projection, candidate periods below N, and room for
  // the pair window plus one fp64 staging buffer in LDS.
  const bool pair = pair_eligible(c, dtype, N, num, min_length, max_length, flags) && !gwin1 && !gwin2;
  PH_TRY(prepare_plan(c, min_length, max_length, &plan, &n_pass, 4, false, pair));
  Stage st(c, flags);
  const void* dx;
  void *dper, *dpow, *dbases, *dstat;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, periods, (size_t)W * num * sizeof(uint32_t), &dper));
  PH_TRY(st.out(B_OUT1, powers, (size_t)W * num * sizeof(double), &dpow));
  PH_TRY(st.out(B_OUT2, bases, (size_t)W * num * N * sz, &dbases));
  PH_TRY(st.out(B_OUT3, status, (size_t)W * sizeof(int32_t), &dstat));
  void* dsweeps;
  PH_TRY(st.out(B_OUT4, n_sweeps, (size_t)W * sizeof(int32_t), &dsweeps));
  PH_TRY(ensure(c, c->buf[B_WS0], (size_t)W * sizeof(double)));
  double* dnorm = static_cast<double*>(c->buf[B_WS0].p);
  // compact basis rows between the two kernels: the first p elements of every row (ph_kernels.h, step 2);
  // 16-byte aligned rows of a whole number of 128-element LDS-DMA pieces
  const int row_stride = (max_length + 127) & ~127;
  PH_TRY(ensure(c, c->buf[B_WS1], (size_t)W * num * row_stride * sz));
  void* drows = c->buf[B_WS1].p;
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const int max_iters = 12 * (P + num) + 64;
  const dim3 grid((unsigned)W);
  const size_t lds_pair = pair_lds_bytes(N, num, P);
  if (pair) {
    const size_t gstride = ph::win_stride((size_t)N);
    PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * gstride * sizeof(double)));
    auto kernel = ph::k_mbest_step1_pair;
    PH_TRY(allow_lds(kernel, lds_pair));
    ProfScope ps_(c, "k_mbest_step1");
    hipLaunchKernelGGL(kernel, dim3((unsigned)((W + 1) / 2)), dim3(1024), lds_pair, c->stream, (const double*)dx, (int)W, N,
                       num, min_length, max_length, gamma, geom, static_cast<const ph::PGeomF*>(c->geomf.p), plan, n_pass,
                       static_cast<double*>(c->buf[B_GWIN].p), max_iters, (uint32_t*)dper, (double*)dpow, (double*)drows,
                       row_stride, dnorm, (int*)dstat, (int*)dsweeps);
  } else
  PH_TRY(dispatch(dtype, !gwin1, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_mbest_step1<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds1));
    ProfScope ps_(c, "k_mbest_step1");
    // 16 wavefronts per window: two workgroups per CU at N = 4096 (-2.4 % against four of 8 waves),
    // and long windows, which leave room for one or two workgroups per CU, still fill the SIMDs
    // (N = 8192: -15 %, N = 16384: -23 %)
    // (short windows, N < 3072, keep 8: four workgroups per CU there)
    const int block1 = c->step1_block ? c->step1_block
                       : (c->sweep_block == ph::kBlockWide && N >= 3072) ? 1024
                                                                         : c->sweep_block;
    hipLaunchKernelGGL(kernel, grid, dim3(block1), lds1, c->stream, (const T*)dx, N, num, min_length,
                       max_length, gamma, kflags, tb, geom, plan, n_pass, (T*)gbuf1, (T*)gwin1, max_iters,
                       (uint32_t*)dper, (double*)dpow, (T*)drows, row_stride, dnorm, (int*)dstat, (int*)dsweeps, small_means);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_mbest_step1"));
  PH_TRY(dispatch(dtype, !gwin2, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_mbest_step2<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds2));
    ProfScope ps_(c, "k_mbest_step2");
    hipLaunchKernelGGL(kernel, grid, dim3(general ? kBlock : c->sweep_block), lds2, c->stream, N, num, gamma,
                       max_length, kflags, tb, geom,
                       max_fac, (T*)gbuf2, (T*)gwin2, (uint32_t*)dper, (double*)dpow, (T*)dbases, dnorm,
                       (const int*)dstat, (T*)drows, row_stride);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_mbest_step2"));
  return st.finish();
}

// ----------------------------------------------------------------------------- small_to_large
int ph_small_to_large(ph_ctx* c, const void* x, int dtype, int64_t W, int N, double thresh, int n_periods,
                      const int32_t* orth_off, const int32_t* orth_q, int table_max_p, unsigned flags, int cap,
                      int32_t* counts, int32_t* periods, double* powers, void* bases, int32_t* status) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!counts || !periods || !powers || !status) return fail(PH_E_ARG, "output pointer is NULL");
  if (cap < 1) return fail(PH_E_ARG, "cap=%d must be >= 1", cap);
  if (n_periods < 0) n_periods = N / 2;  // Periods.py:271-272
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const bool general = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  size_t lds = carve_bytes(N + kPad, sz) + (general ? carve_bytes(N, sz) : carve_bytes(kBlockWide, sz)) + carve_bytes(kRedDoubles, 8) +
               carve_bytes(ph::kS2LBatch, 8) + carve_bytes(4, 4);
  void* gbuf;
  PH_TRY(place_second_buffer(c, &lds, general, (size_t)N * sz, W, &gbuf));
  void* gwin;
  PH_TRY(place_window(c, &lds, N + kPad, sz, W, &gwin));
  PH_TRY(check_lds(c, lds, N, "ph_small_to_large"));
  const ph::PGeom* geom;
  PH_TRY(prepare_geom(c, N, std::max(n_periods, 2), &geom));
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, flags, orth_off, orth_q, table_max_p, std::max(n_periods, 1), &tb));
  Stage st(c, flags);
  const void* dx;
  void *dcnt, *dper, *dpow, *dbases, *dstat;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, counts, (size_t)W * sizeof(int32_t), &dcnt));
  PH_TRY(st.out(B_OUT1, periods, (size_t)W * cap * sizeof(int32_t), &dper));
  PH_TRY(st.out(B_OUT2, powers, (size_t)W * cap * sizeof(double), &dpow));
  PH_TRY(st.out(B_OUT3, bases, (size_t)W * cap * N * sz, &dbases));
  PH_TRY(st.out(B_OUT4, status, (size_t)W * sizeof(int32_t), &dstat));
  PH_HIP(hipMemsetAsync(dper, 0, (size_t)W * cap * sizeof(int32_t), c->stream));
  PH_HIP(hipMemsetAsync(dpow, 0, (size_t)W * cap * sizeof(double), c->stream));
  PH_TRY(ensure(c, c->buf[B_GEN0], 256));
  int* dmax = static_cast<int*>(c->buf[B_GEN0].p);  // largest count of the batch (device word)
  PH_HIP(hipMemsetAsync(dmax, 0, 2 * sizeof(int), c->stream));  // [1]: pair counter of the persistent pair kernel
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const dim3 grid((unsigned)W);
  // Window-pair screen (ph_s2l.h): fp64 windows, plain projection, candidate periods below N.  LDS: the pair window and
  // one fp64 staging buffer, two 16-wave workgroups per CU; the fp64 residuals live in an HBM workspace.
  const size_t lds_pair = ph::s2l_pair_lds_bytes(N);
  const bool pair_ok = c->s2l_pair && dtype == PH_F64 && !general && !gwin && n_periods < N && n_periods >= 2 &&
                       c->sweep_block >= 512;
  const bool pair = pair_ok && lds_pair <= (size_t)c->lds_limit;
  if (pair) {
    const size_t gstride = ph::win_stride((size_t)N);
    PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * gstride * sizeof(double)));
    auto kernel = ph::k_small_to_large_pair;
    PH_TRY(allow_lds(kernel, lds_pair));
    ProfScope ps_(c, "k_small_to_large");
    // persistent workgroups: as many as are resident at once (LDS and the 32 wavefronts of a CU), pairs from a counter
    const int64_t npairs = (W + 1) / 2;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->lds_limit / lds_pair, 2048 / (size_t)c->s2l_block));
    const unsigned grid_pairs = (unsigned)std::min<int64_t>(npairs, (int64_t)c->num_cu * per_cu);
    hipLaunchKernelGGL(kernel, dim3(grid_pairs), dim3(c->s2l_block), lds_pair, c->stream, (const double*)dx, (int)W, N,
                       thresh, n_periods, static_cast<const ph::PGeomF*>(c->geomf.p), static_cast<const float*>(c->kapf.p),
                       static_cast<double*>(c->buf[B_GWIN].p), cap, (int*)dcnt, (int*)dper, (double*)dpow, (double*)dbases,
                       (int*)dstat, dmax, dmax + 1);
  } else
  PH_TRY(dispatch(dtype, !gwin, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_small_to_large<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_small_to_large");
    hipLaunchKernelGGL(kernel, grid, dim3(c->sweep_block), lds, c->stream, (const T*)dx, N, thresh, n_periods, kflags,
                       tb, geom, (T*)gbuf, (T*)gwin, cap, (int*)dcnt, (int*)dper, (double*)dpow, (T*)dbases,
                       (int*)dstat, dmax);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_small_to_large"));
  PH_TRY(st.finish());
  // Capacity overflow must be impossible to miss: host-pointer calls are synchronous anyway; a
  // device-pointer call reads the batch maximum back (one word, one stream synchronisation) unless
  // the caller opted out with PH_FLAG_NOSYNC and checks `status` / `counts` itself.
  if (!(flags & PH_FLAG_DEVICE) || !(flags & PH_FLAG_NOSYNC)) {
    int worst = 0;
    PH_HIP(hipMemcpyAsync(&worst, dmax, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PH_HIP(hipStreamSynchronize(c->stream));
    if (worst > cap) return fail(PH_E_CAP, "a window accepted %d periods, cap is %d", worst, cap);
  }
  return PH_OK;
}

// ----------------------------------------------------------------------------- best_correlation
int ph_best_correlation(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int num, int max_length,
                        double ratio, const int32_t* orth_off, const int32_t* orth_q, int table_max_p,
                        unsigned flags, uint32_t* periods, double* norms, void* bases, int32_t* status) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!periods || !norms || !bases || !status) return fail(PH_E_ARG, "output pointer is NULL");
  if (num < 1) return fail(PH_E_ARG, "num=%d must be >= 1", num);
  if (max_length < 0) max_length = N / 3;  // Periods.py:311-312
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const bool general = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  size_t lds = carve_bytes(N + kPad, sz) + (general ? carve_bytes(N, sz) : 0) + carve_bytes(kRedDoubles, 8) +
               carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4) + carve_bytes(ph::kCandCap, 4) +
               carve_bytes(ph::kCandCap, 8) + carve_bytes(1, sizeof(ph::CandCtl));
  void* gbuf;
  PH_TRY(place_second_buffer(c, &lds, general, (size_t)N * sz, W, &gbuf));
  void* gwin;
  PH_TRY(place_window(c, &lds, N + kPad, sz, W, &gwin));
  PH_TRY(check_lds(c, lds, N, "ph_best_correlation"));
  const ph::PassPlan* plan = nullptr;
  int n_pass = 0;
  // Window-pair screen (k_best_correlation_pair): fp64 windows, plain projection, pair window + staging buffer in LDS
  const size_t lds_pair = 2 * carve_bytes(N + kPad, 8) + carve_bytes(kRedDoubles, 8) + carve_bytes(kMaxWaves, 8) +
                          carve_bytes(kMaxWaves, 4) + carve_bytes(2 * ph::kPairListCap, 4) + carve_bytes(8, 4) +
                          carve_bytes(10, 8);
  const bool pair = c->bc_pair && dtype == PH_F64 && !general && !gwin && max_length <= N && max_length - 1 >= 2 &&
                    lds_pair <= (size_t)c->lds_limit;
  if (max_length - 1 >= 2) PH_TRY(prepare_plan(c, 2, max_length - 1, &plan, &n_pass, 4, true, pair));
  const ph::PGeom* geom;
  PH_TRY(prepare_geom(c, N, std::max(max_length, 2), &geom));
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, flags, orth_off, orth_q, table_max_p, std::max(max_length, 1), &tb));
  Stage st(c, flags);
  const void* dx;
  void *dper, *dnrm, *dbases, *dstat;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, periods, (size_t)W * num * sizeof(uint32_t), &dper));
  PH_TRY(st.out(B_OUT1, norms, (size_t)W * num * sizeof(double), &dnrm));
  PH_TRY(st.out(B_OUT2, bases, (size_t)W * num * N * sz, &dbases));
  PH_TRY(st.out(B_OUT3, status, (size_t)W * sizeof(int32_t), &dstat));
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const dim3 grid((unsigned)W);
  if (pair) {
    const size_t gstride = ph::win_stride((size_t)N);
    PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * gstride * sizeof(double)));
    auto kernel = ph::k_best_correlation_pair;
    PH_TRY(allow_lds(kernel, lds_pair));
    ProfScope ps_(c, "k_best_correlation");
    hipLaunchKernelGGL(kernel, dim3((unsigned)((W + 1) / 2)), dim3(1024), lds_pair, c->stream, (const double*)dx, (int)W, N,
                       num, max_length, ratio, geom, static_cast<const ph::PGeomF*>(c->geomf.p), plan, n_pass,
                       static_cast<double*>(c->buf[B_GWIN].p), (uint32_t*)dper, (double*)dnrm, (double*)dbases, (int*)dstat);
  } else
  PH_TRY(dispatch(dtype, !gwin, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_best_correlation<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_best_correlation");
    hipLaunchKernelGGL(kernel, grid, dim3(c->sweep_block), lds, c->stream, (const T*)dx, N, num, max_length, ratio,
                       kflags, tb, geom, plan, n_pass, (T*)gbuf, (T*)gwin, (uint32_t*)dper, (double*)dnrm, (T*)dbases,
                       (int*)dstat);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_best_correlation"));
  return st.finish();
}

// ----------------------------------------------------------------------------- best_frequency
int ph_best_frequency(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int win_size, int num,
                      const int32_t* orth_off, const int32_t* orth_q, int table_max_p, unsigned flags,
                      uint32_t* periods, double* powers, void* bases, int32_t* status) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!periods || !powers || !bases || !status) return fail(PH_E_ARG, "output pointer is NULL");
  if (num < 1) return fail(PH_E_ARG, "num=%d must be >= 1", num);
  const int L = win_size < 1 ? N : win_size;  // Periods.py:381-382
  if (L < 2 || L > (1 << 24)) return fail(PH_E_ARG, "win_size=%d out of range", L);
  if (W > 65535) return fail(PH_E_ARG, "ph_best_frequency: W=%lld exceeds 65535 windows per call", (long long)W);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  size_t lds = 2 * carve_bytes(N, sz) + carve_bytes(kRedDoubles, 8);  // update kernel
  void* gbuf;
  PH_TRY(place_second_buffer(c, &lds, true, (size_t)N * sz, W, &gbuf));
  // windows longer than the LDS: both kernels work on the residual where it lives (HBM workspace)
  const bool lds_window = lds <= (size_t)c->lds_limit;
  if (!lds_window) lds -= carve_bytes(N, sz);
  PH_TRY(check_lds(c, lds, N, "ph_best_frequency"));
  const size_t lds_spec = (lds_window ? carve_bytes(N, sz) : 0) + carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4);
  PH_TRY(check_lds(c, lds_spec, N, "ph_best_frequency"));
  // power-of-two win_size whose complex work array fits the LDS: in-LDS FFT, one record per window
  const size_t lds_fft = 2 * carve_bytes(L, 8) + carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4);
  const bool use_fft = (L & (L - 1)) == 0 && L >= 4 && lds_fft <= (size_t)c->lds_limit && !std::getenv("PH_BF_DIRECT");
  int logL = 0;
  while ((1 << logL) < L) ++logL;
  // any other win_size: Bluestein's chirp convolution on two FFTs of size M >= min(N, L) + L / 2 + 1, if that fits
  const int M0 = std::min(N, L);
  int logM = 0;
  while ((1LL << logM) < (long long)M0 + L / 2 + 1) ++logM;
  const int M = 1 << logM;
  const size_t lds_chirp = 2 * carve_bytes(M, 8) + carve_bytes(kMaxWaves, 8) + carve_bytes(kMaxWaves, 4);
  const bool use_chirp = !use_fft && L >= 4 && logM <= 20 && lds_chirp <= (size_t)c->lds_limit && !std::getenv("PH_BF_DIRECT");
  const int nchunk = (use_fft || use_chirp) ? 1 : (L / 2 + 1 + ph::kBfBlock - 1) / ph::kBfBlock;
  const int wlen = std::max(M0, L / 2 + 1);  // chirp entries the kernel and the wrapped filter need
  if (use_chirp && (c->bs_L != L || c->bs_M0 != M0)) {
    const long double pi = 3.141592653589793238462643383279L;
    std::vector<double> tab(2 * ((size_t)M + wlen + M));  // [twiddles M | chirp wlen | B M] as (re, im) pairs
    double* twm = tab.data();
    double* chp = twm + 2 * (size_t)M;
    double* bf = chp + 2 * (size_t)wlen;
    for (int k = 0; k < M; ++k) {
      const long double a = 2.0L * pi * (long double)k / (long double)M;
      twm[2 * (size_t)k] = (double)std::cos(a);
      twm[2 * (size_t)k + 1] = (double)std::sin(a);
    }
    for (int n = 0; n < wlen; ++n) {  // n^2 mod 2L keeps the angle exact and small
      const long double a = pi * (long double)(((long long)n * n) % (2LL * L)) / (long double)L;
      chp[2 * (size_t)n] = (double)std::cos(a);
      chp[2 * (size_t)n + 1] = (double)std::sin(a);
    }
    // B = FFT_M of the wrapped conj(w): b[m] = exp(+i pi m^2 / L) for m in [0, L/2] and at M - m for m in [1, M0)
    std::vector<long double> br((size_t)M, 0.0L), bi((size_t)M, 0.0L);
    for (int m = 0; m <= L / 2; ++m) {
      br[m] = chp[2 * (size_t)m];
      bi[m] = chp[2 * (size_t)m + 1];
    }
    for (int m = 1; m < M0; ++m) {
      br[M - m] = chp[2 * (size_t)m];
      bi[M - m] = chp[2 * (size_t)m + 1];
    }
    for (int i = 1, j = 0; i < M; ++i) {  // bit reversal, then iterative radix-2 (host, long double)
      int bit = M >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) {
        std::swap(br[i], br[j]);
        std::swap(bi[i], bi[j]);
      }
    }
    for (int len = 2; len <= M; len <<= 1) {
      const int half = len >> 1;
      for (int j = 0; j < half; ++j) {
        const long double a = -2.0L * pi * (long double)j / (long double)len;
        const long double wr = std::cos(a), wi = std::sin(a);
        for (int i = j; i < M; i += len) {
          const long double xr = br[i + half] * wr - bi[i + half] * wi, xi = br[i + half] * wi + bi[i + half] * wr;
          br[i + half] = br[i] - xr;
          bi[i + half] = bi[i] - xi;
          br[i] += xr;
          bi[i] += xi;
        }
      }
    }
    for (int k = 0; k < M; ++k) {
      bf[2 * (size_t)k] = (double)br[k];
      bf[2 * (size_t)k + 1] = (double)bi[k];
    }
    PH_HIP(hipStreamSynchronize(c->stream));
    PH_TRY(ensure(c, c->bs_tab, tab.size() * sizeof(double)));
    PH_HIP(hipMemcpyAsync(c->bs_tab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PH_HIP(hipStreamSynchronize(c->stream));
    c->bs_L = L;
    c->bs_M0 = M0;
  }
  ph::Tables tb{};
  PH_TRY(prepare_orth(c, flags, orth_off, orth_q, table_max_p, 2 * L, &tb));
  if (c->twid_len != L) {  // twiddles in float64; k / L is an exact fraction of a turn
    std::vector<double> tw(2 * (size_t)L);
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int k = 0; k < L; ++k) {
      const long double a = two_pi * (long double)k / (long double)L;
      tw[2 * (size_t)k] = (double)std::cos(a);
      tw[2 * (size_t)k + 1] = (double)std::sin(a);
    }
    PH_HIP(hipStreamSynchronize(c->stream));
    PH_TRY(ensure(c, c->twid, tw.size() * sizeof(double)));
    PH_HIP(hipMemcpyAsync(c->twid.p, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PH_HIP(hipStreamSynchronize(c->stream));
    c->twid_len = L;
  }
  Stage st(c, flags);
  const void* dx;
  void *dper, *dpow, *dbases, *dstat;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, periods, (size_t)W * num * sizeof(uint32_t), &dper));
  PH_TRY(st.out(B_OUT1, powers, (size_t)W * num * sizeof(double), &dpow));
  PH_TRY(st.out(B_OUT2, bases, (size_t)W * num * N * sz, &dbases));
  PH_TRY(st.out(B_OUT3, status, (size_t)W * sizeof(int32_t), &dstat));
  // workspace: the residual, per-chunk spectral maxima, ||data||
  PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * N * sz));
  PH_TRY(ensure(c, c->buf[B_WS0], (size_t)W * nchunk * sizeof(double) + (size_t)W * sizeof(double)));
  PH_TRY(ensure(c, c->buf[B_WS1], (size_t)W * nchunk * sizeof(int)));
  void* dres = c->buf[B_GWIN].p;
  double* dpart = static_cast<double*>(c->buf[B_WS0].p);
  double* dnrm = dpart + (size_t)W * nchunk;
  int* dpartk = static_cast<int*>(c->buf[B_WS1].p);
  PH_HIP(hipMemcpyAsync(dres, dx, (size_t)W * N * sz, hipMemcpyDeviceToDevice, c->stream));
  PH_HIP(hipMemsetAsync(dstat, 0, (size_t)W * sizeof(int32_t), c->stream));
  const unsigned kflags = flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH);
  const dim3 grid_s((unsigned)nchunk, (unsigned)W), grid_u((unsigned)W);
  PH_TRY(dispatch(dtype, lds_window, [&](auto t, auto lw) {
    using T = decltype(t);
    constexpr bool LW = decltype(lw)::value;
    PH_TRY(allow_lds(ph::k_bf_spectrum<T, LW>, lds_spec));
    if (use_fft) PH_TRY(allow_lds(ph::k_bf_fft<T>, lds_fft));
    if (use_chirp) PH_TRY(allow_lds(ph::k_bf_chirp<T>, lds_chirp));
    PH_TRY(allow_lds(ph::k_bf_update<T, LW>, lds));
    for (int it = 0; it < num; ++it) {
      if (use_fft) {
        ProfScope ps_(c, "k_bf_fft");
        hipLaunchKernelGGL((ph::k_bf_fft<T>), grid_u, dim3(kBlockWide), lds_fft, c->stream, (const T*)dres, N, L, logL,
                           (const double2*)c->twid.p, (const int*)dstat, dpart, dpartk);
      } else if (use_chirp) {
        const double2* twm = (const double2*)c->bs_tab.p;
        ProfScope ps_(c, "k_bf_chirp");
        hipLaunchKernelGGL((ph::k_bf_chirp<T>), grid_u, dim3(kBlockWide), lds_chirp, c->stream, (const T*)dres, N, L, M, logM,
                           twm, twm + M, twm + M + wlen, (const int*)dstat, dpart, dpartk);
      } else {
        ProfScope ps_(c, "k_bf_spectrum");
        hipLaunchKernelGGL((ph::k_bf_spectrum<T, LW>), grid_s, dim3(ph::kBfBlock), lds_spec, c->stream, (const T*)dres, N, L,
                           (const double2*)c->twid.p, (const int*)dstat, dpart, dpartk);
      }
      PH_TRY(launch_check("k_bf_spectrum"));
      {
        ProfScope ps_(c, "k_bf_update");
        hipLaunchKernelGGL((ph::k_bf_update<T, LW>), grid_u, dim3(kBlockWide), lds, c->stream, (T*)dres, N, L, num, it,
                           kflags, tb, (T*)gbuf, nchunk, (const double*)dpart, (const int*)dpartk, dnrm,
                           (uint32_t*)dper, (double*)dpow, (T*)dbases, (int*)dstat);
      }
      PH_TRY(launch_check("k_bf_update"));
    }
    return (int)PH_OK;
  }));
  return st.finish();
}

// ----------------------------------------------------------------------------- Ramanujan
int ph_ramanujan_norms(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int q_lo, int q_hi,
                       unsigned flags, double* out) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!out) return fail(PH_E_ARG, "out is NULL");
  if (q_lo < 1 || q_hi < 1) return fail(PH_E_ARG, "need q_lo, q_hi >= 1 (got %d, %d)", q_lo, q_hi);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  // Per wavefront: strip A (q_hi doubles, the root fold) and strip B (q_hi / 2, one child).  As many
  // wavefronts as the LDS left by the window allows, at most 16; when fewer than four fit beside an
  // LDS-resident window (the reference's default range q_hi = N / 3 on a long window) the window moves
  // to the HBM workspace and the strips get the whole LDS.
  const size_t strip = carve_bytes((size_t)q_hi, 8) + carve_bytes((size_t)std::max(1, q_hi / 2), 8);
  const size_t win_bytes = carve_bytes(N + kPad, sz);
  auto waves_for = [&](size_t room) { return (int)std::min<size_t>(ph::kRamMaxWaves, room / strip); };
  const size_t limit = (size_t)c->lds_limit;
  int nw = win_bytes < limit ? waves_for(limit - win_bytes) : 0;
  int pad = kPad;
  // The zeroed pad behind the window serves the row-split fold of roots below 64 (q_hi < 128) only; a fold of a root
  // >= 64 merely reads up to 255 elements past the window for lanes whose sums it discards.  Without the pad those reads
  // land in the strips -- and at config 3 window + 16 x 6 KB of strips are exactly the 160 KB of the CU.
  const size_t win_bare = carve_bytes(N, sz);
  if (q_hi >= 128 && nw >= 2 && nw < ph::kRamMaxWaves && waves_for(limit - win_bare) > nw) {
    nw = waves_for(limit - win_bare);
    pad = 0;
  }
  void* gwin = nullptr;
  if (nw < 4) {
    const int nw_hbm = waves_for(limit);
    if (nw_hbm > nw) {
      nw = nw_hbm;
      pad = kPad;
      PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * ph::win_stride(N + kPad) * sz));
      gwin = c->buf[B_GWIN].p;
    }
  }
  if (nw < 1)
    return fail(PH_E_ARG, "ph_ramanujan_norms: q_hi=%d needs %zu B of LDS per wavefront, device limit is %d B", q_hi, strip,
                c->lds_limit);
  const size_t lds = (gwin ? 0 : pad ? win_bytes : win_bare) + carve_bytes((size_t)nw * q_hi, 8) +
                     carve_bytes((size_t)nw * std::max(1, q_hi / 2), 8);
  PH_TRY(check_lds(c, lds, N, "ph_ramanujan_norms"));
  // One 128-byte record per period (ph::RamJob): the factors (I - P_d) of its projector (d = q / r for each prime
  // r | q, with 1 / r and the row-split geometry of a coset count below 64), the scale (q / phi(q))^2 and the
  // geometry of its fold of the window.
  if (q_hi >= 30030 || q_hi > 0xffff) return fail(PH_E_ARG, "ph_ramanujan_norms: q_hi=%d is beyond the record format", q_hi);
  auto small_geom = [](int d) { return d >= 1 && d < 64 ? (64 / d) | (((65536 + d - 1) / d) << 8) : 0; };
  std::vector<ph::RamJob> job((size_t)q_hi + 1);
  {
    std::vector<int32_t> phi(q_hi + 1);
    for (int i = 0; i <= q_hi; ++i) phi[i] = i;
    std::vector<std::vector<int32_t>> primes_of(q_hi + 1);
    std::vector<char> comp(q_hi + 1, 0);
    for (int i = 2; i <= q_hi; ++i) {
      if (comp[i]) continue;
      for (int j = i; j <= q_hi; j += i) {
        comp[j] = j > i;
        phi[j] -= phi[j] / i;
        primes_of[j].push_back(i);
      }
    }
    for (int q = 1; q <= q_hi; ++q) {
      ph::RamJob& J = job[q];
      std::memset(&J, 0, sizeof(J));
      J.q = q;
      J.k = 1;
      J.rows = (N + q - 1) / q;            // as PGeom: residues j < nfull own `rows` samples, the others rows - 1
      J.nfull = N - (J.rows - 1) * q;
      J.sm = small_geom(q);
      const double sc = (double)q / (double)phi[q];
      J.scale2 = sc * sc;
      int ns = 0;
      for (int r : primes_of[q]) {
        J.st[ns].dr = (q / r) | (r << 16);
        J.st[ns].sm = small_geom(q / r);
        J.st[ns].inv_r = 1.0 / (double)r;
        ++ns;
      }
      J.flags = ns << 8;
    }
  }
  // Root plan: roots are the periods of (q_hi/2, q_hi]; every wanted q <= q_hi/2 becomes the child of one
  // of its multiples there (the least loaded one; children cost a strip fold and a filter, O(Q + q)).
  // Roots are dealt to the wavefronts in order of decreasing work.
  std::vector<ph::RamJob> tab;  // root records, then the children
  int n_root = 0;
  if (q_lo <= q_hi) {
    const int half = q_hi / 2;
    std::vector<std::vector<int32_t>> kids((size_t)q_hi + 1);
    std::vector<int64_t> load((size_t)q_hi + 1, 0);
    for (int Q = half + 1; Q <= q_hi; ++Q) load[Q] = (int64_t)N + Q;
    for (int q = std::min(half, q_hi); q >= q_lo; --q) {
      int best = 0;
      for (int Q = ((half / q) + 1) * q; Q <= q_hi; Q += q)
        if (!best || load[Q] < load[best]) best = Q;
      kids[best].push_back(q);
      load[best] += 2 * (int64_t)best + 4 * q;
    }
    std::vector<int> order;
    for (int Q = half + 1; Q <= q_hi; ++Q)
      if (Q >= q_lo || !kids[Q].empty()) order.push_back(Q);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] > load[b]; });
    n_root = (int)order.size();
    tab.resize((size_t)n_root);
    for (int i = 0; i < n_root; ++i) {
      const int Q = order[i];
      ph::RamJob R = job[Q];
      R.c0 = (int32_t)(tab.size() - (size_t)n_root);
      for (int q : kids[Q]) {
        ph::RamJob C = job[q];
        C.k = Q / q;
        tab.push_back(C);
      }
      R.c1 = (int32_t)(tab.size() - (size_t)n_root);
      R.flags |= Q >= q_lo ? 1 : 0;
      tab[i] = R;
    }
  }
  if (tab.empty()) tab.push_back(ph::RamJob{});
  const int* d_tab;  // the records travel as raw int32 words through a cached table slot
  PH_TRY(upload_table(c, T_AUX2, reinterpret_cast<const int32_t*>(tab.data()), tab.size() * (sizeof(ph::RamJob) / 4), &d_tab));
  Stage st(c, flags);
  const void* dx;
  void* dout;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * (q_hi + 1) * sizeof(double), &dout));
  PH_HIP(hipMemsetAsync(dout, 0, (size_t)W * (q_hi + 1) * sizeof(double), c->stream));
  const dim3 grid((unsigned)W);
  if (n_root > 0) {
    const ph::RamJob* d_roots = reinterpret_cast<const ph::RamJob*>(d_tab);
    PH_TRY(dispatch(dtype, !gwin, [&](auto t, auto lw) {
      using T = decltype(t);
      auto kernel = ph::k_ramanujan<T, decltype(lw)::value>;
      PH_TRY(allow_lds(kernel, lds));
      ProfScope ps_(c, "k_ramanujan");
      hipLaunchKernelGGL(kernel, grid, dim3(nw * 64), lds, c->stream, (const T*)dx, N, q_hi, d_roots, n_root, d_roots + n_root,
                         (T*)gwin, pad, (double*)dout);
      return (int)PH_OK;
    }));
    PH_TRY(launch_check("k_ramanujan"));
  }
  return st.finish();
}

// ----------------------------------------------------------------------------- QOPeriods blocks
static int qo_tables(ph_ctx* c, const int32_t* p_list, const int32_t* keep, int n_p, const int** d_p,
                     const int** d_keep, const int** d_off, int* stride) {
  if (!p_list || !keep || n_p < 1) return fail(PH_E_ARG, "p_list/keep NULL or n_p < 1");
  std::vector<int32_t> off(n_p + 1, 0);
  for (int k = 0; k < n_p; ++k) {
    if (p_list[k] < 1 || keep[k] < 0 || keep[k] > p_list[k])
      return fail(PH_E_ARG, "need p >= 1 and 0 <= keep <= p at entry %d", k);
    off[k + 1] = off[k] + keep[k];
  }
  *stride = off[n_p];
  if (*stride < 1) return fail(PH_E_ARG, "sum(keep) must be >= 1");
  PH_TRY(upload_table(c, T_PLIST, p_list, n_p, d_p));
  PH_TRY(upload_table(c, T_AUX0, keep, n_p, d_keep));
  PH_TRY(upload_table(c, T_AUX1, off.data(), off.size(), d_off));
  return PH_OK;
}

int ph_fold_sums(ph_ctx* c, const void* x, int dtype, int64_t W, int N, const int32_t* p_list,
                 const int32_t* keep, int n_p, unsigned flags, double* out) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!out) return fail(PH_E_ARG, "out is NULL");
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const bool lds_window = carve_bytes(N, sz) <= (size_t)c->lds_limit;  // longer windows are folded from HBM / L2
  const size_t lds = lds_window ? carve_bytes(N, sz) : 0;
  const int *d_p, *d_keep, *d_off;
  int stride;
  PH_TRY(qo_tables(c, p_list, keep, n_p, &d_p, &d_keep, &d_off, &stride));
  Stage st(c, flags);
  const void* dx;
  void* dout;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * stride * sizeof(double), &dout));
  const dim3 grid((unsigned)W);
  PH_TRY(dispatch(dtype, lds_window, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_fold_sums<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_fold_sums");
    hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, c->stream, (const T*)dx, N, d_p, d_keep, d_off, n_p, stride,
                       (double*)dout);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_fold_sums"));
  return st.finish();
}

int ph_tile_sum(ph_ctx* c, const double* wts, int64_t W, int N, const int32_t* p_list, const int32_t* keep,
                int n_p, int dtype, unsigned flags, void* out) {
  PH_TRY(check_common(c, wts, dtype, W, N));
  if (!out) return fail(PH_E_ARG, "out is NULL");
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  const int *d_p, *d_keep, *d_off;
  int stride;
  PH_TRY(qo_tables(c, p_list, keep, n_p, &d_p, &d_keep, &d_off, &stride));
  const size_t lds = carve_bytes(stride, 8);
  PH_TRY(check_lds(c, lds, N, "ph_tile_sum"));
  Stage st(c, flags);
  const void* dw;
  void* dout;
  PH_TRY(st.in(wts, (size_t)W * stride * sizeof(double), &dw));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * N * sz, &dout));
  const dim3 grid((unsigned)W);
  if (dtype == PH_F64) {
    PH_TRY(allow_lds(ph::k_tile_sum<double>, lds));
    {
      ProfScope ps_(c, "k_tile_sum");
      hipLaunchKernelGGL(ph::k_tile_sum<double>, grid, dim3(kBlock), lds, c->stream, (const double*)dw, N, d_p,
                         d_keep, d_off, n_p, stride, (double*)dout);
    }
  } else {
    PH_TRY(allow_lds(ph::k_tile_sum<float>, lds));
    {
      ProfScope ps_(c, "k_tile_sum");
      hipLaunchKernelGGL(ph::k_tile_sum<float>, grid, dim3(kBlock), lds, c->stream, (const double*)dw, N, d_p,
                         d_keep, d_off, n_p, stride, (float*)dout);
    }
  }
  PH_TRY(launch_check("k_tile_sum"));
  return st.finish();
}

// ----------------------------------------------------------------------------- periodic_norm
int ph_periodic_norm(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int p, unsigned flags,
                     double* out) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!out) return fail(PH_E_ARG, "out is NULL");
  if (p < 0) return fail(PH_E_ARG, "p=%d must be >= 0 (0 = no period normalisation)", p);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  Stage st(c, flags);
  const void* dx;
  void* dout;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, out, (size_t)W * sizeof(double), &dout));
  const dim3 grid((unsigned)W);
  if (dtype == PH_F64)
    {
      ProfScope ps_(c, "k_periodic_norm");
      hipLaunchKernelGGL(ph::k_periodic_norm<double>, grid, dim3(kBlock), 0, c->stream, (const double*)dx, N, p,
                         (double*)dout);
    }
  else
    {
      ProfScope ps_(c, "k_periodic_norm");
      hipLaunchKernelGGL(ph::k_periodic_norm<float>, grid, dim3(kBlock), 0, c->stream, (const float*)dx, N, p,
                         (double*)dout);
    }
  PH_TRY(launch_check("k_periodic_norm"));
  return st.finish();
}

// ----------------------------------------------------------------------------- dictionary project
int ph_dict_project(ph_ctx* c, const double* x, const double* basis, int rows, int N, unsigned flags,
                    float* out) {
  if (!c || !x || !basis || !out) return fail(PH_E_ARG, "NULL argument");
  if (rows < 1 || N < 1) return fail(PH_E_ARG, "rows=%d, N=%d must be >= 1", rows, N);
  PH_HIP(hipSetDevice(c->device));
  const bool device = flags & PH_FLAG_DEVICE;
  const double *dx = x, *db = basis;
  float* dout = out;
  if (!device) {
    PH_TRY(ensure(c, c->buf[B_IN], (size_t)N * 8));
    PH_TRY(ensure(c, c->buf[B_WS1], (size_t)rows * N * 8));
    PH_TRY(ensure(c, c->buf[B_OUT0], (size_t)rows * N * 4));
    PH_HIP(hipMemcpyAsync(c->buf[B_IN].p, x, (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
    PH_HIP(hipMemcpyAsync(c->buf[B_WS1].p, basis, (size_t)rows * N * 8, hipMemcpyHostToDevice, c->stream));
    dx = (const double*)c->buf[B_IN].p;
    db = (const double*)c->buf[B_WS1].p;
    dout = (float*)c->buf[B_OUT0].p;
  }
  {
    ProfScope ps_(c, "k_dict_project");
    hipLaunchKernelGGL(ph::k_dict_project, dim3((unsigned)rows), dim3(kBlock), 0, c->stream, dx, db, N, dout);
  }
  PH_TRY(launch_check("k_dict_project"));
  if (!device) {
    PH_HIP(hipMemcpyAsync(out, dout, (size_t)rows * N * 4, hipMemcpyDeviceToHost, c->stream));
    PH_HIP(hipStreamSynchronize(c->stream));
  }
  return PH_OK;
}

// ----------------------------------------------------------------------------- QOPeriods.find_periods
int ph_qo_find_periods(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int num, double thresh,
                       int min_length, int max_length, int kcap, unsigned flags, uint32_t* periods,
                       double* norms, int32_t* keeps, int32_t* counts, double* weights, void* residual,
                       int32_t* status) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!periods || !norms || !keeps || !counts || !weights || !residual || !status)
    return fail(PH_E_ARG, "output pointer is NULL");
  if (flags & (PH_FLAG_TRUNC | PH_FLAG_ORTH))
    return fail(PH_E_UNSUPPORTED, "ph_qo_find_periods implements the plain-projection branch only");
  if (num < 1) return fail(PH_E_ARG, "num=%d must be >= 1", num);
  if (max_length < 0) max_length = N / 3;  // QOPeriods.py:374-375
  if (min_length < 1 || max_length < min_length)
    return fail(PH_E_ARG, "need 1 <= min_length <= max_length (got %d, %d)", min_length, max_length);
  if (kcap < 1 || kcap > 2048) return fail(PH_E_ARG, "kcap=%d must be in [1, 2048]", kcap);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  size_t lds;
  bool lds_window, overlay;
  PH_TRY(qo_lds_layout(c, sz, N, max_length, kcap, &lds, &lds_window, &overlay));
  void* gwin = nullptr;
  if (!lds_window) {
    PH_TRY(ensure(c, c->buf[B_GWIN], (size_t)W * ph::win_stride(N + kPad) * sz));
    gwin = c->buf[B_GWIN].p;
  }
  const ph::PGeom* geom;
  PH_TRY(prepare_geom(c, N, max_length, &geom));
  const ph::PassPlan* plan;
  int n_pass;
  PH_TRY(prepare_plan(c, min_length, max_length, &plan, &n_pass));
  // Euler phi and all divisors of every candidate period (QOPeriods.py:834-838)
  std::vector<int32_t> phi(max_length + 1), off(max_length + 2, 0), dq;
  for (int i = 0; i <= max_length; ++i) phi[i] = i;
  for (int i = 2; i <= max_length; ++i)
    if (phi[i] == i)
      for (int j = i; j <= max_length; j += i) phi[j] -= phi[j] / i;
  for (int q = 0; q <= max_length; ++q) {
    off[q] = (int32_t)dq.size();
    for (int d = 1; q > 0 && d <= q; ++d)
      if (q % d == 0) dq.push_back(d);
  }
  off[max_length + 1] = (int32_t)dq.size();
  if (dq.empty()) dq.push_back(1);
  const int *d_phi, *d_off, *d_dq;
  PH_TRY(upload_table(c, T_AUX0, phi.data(), phi.size(), &d_phi));
  PH_TRY(upload_table(c, T_AUX1, off.data(), off.size(), &d_off));
  PH_TRY(upload_table(c, T_AUX2, dq.data(), dq.size(), &d_dq));
  PH_TRY(ensure(c, c->buf[B_WS1], (size_t)W * 2 * kcap * sizeof(double)));  // last good weights + rhs of every window
  Stage st(c, flags);
  const void* dx;
  void *dper, *dnrm, *dkeep, *dcnt, *dwts, *dres, *dstat;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, periods, (size_t)W * num * sizeof(uint32_t), &dper));
  PH_TRY(st.out(B_OUT1, norms, (size_t)W * num * sizeof(double), &dnrm));
  PH_TRY(st.out(B_OUT2, keeps, (size_t)W * num * sizeof(int32_t), &dkeep));
  PH_TRY(st.out(B_OUT3, counts, (size_t)W * 2 * sizeof(int32_t), &dcnt));
  PH_TRY(st.out(B_OUT4, weights, (size_t)W * kcap * sizeof(double), &dwts));
  PH_TRY(st.out(B_WS0, residual, (size_t)W * N * sz, &dres));
  PH_TRY(st.out(B_GEN0, status, (size_t)W * sizeof(int32_t), &dstat));
  const dim3 grid((unsigned)W);
  PH_TRY(dispatch(dtype, lds_window, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_qo_find<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_qo_find");
    hipLaunchKernelGGL(kernel, grid, dim3(c->qo_block), lds, c->stream, (const T*)dx, N, num, thresh, min_length,
                       max_length, geom, plan, n_pass, d_phi, d_off, d_dq, kcap, overlay ? 1 : 0, (T*)gwin, (double*)c->buf[B_WS1].p,
                       (uint32_t*)dper, (double*)dnrm, (int*)dkeep, (int*)dcnt, (double*)dwts, (T*)dres, (int*)dstat);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_qo_find"));
  return st.finish();
}

int ph_qo_feasible(ph_ctx* c, int dtype, int N, int max_length, int kcap, int* ok) {
  if (!c || !ok) return fail(PH_E_ARG, "NULL argument");
  *ok = 0;
  if (N < 1 || kcap < 1 || kcap > 2048) return PH_OK;
  if (max_length < 0) max_length = N / 3;
  size_t lds;
  bool lds_window, overlay;
  if (qo_lds_layout(c, elem_size(dtype), N, max_length, kcap, &lds, &lds_window, &overlay) == PH_OK) *ok = 1;
  return PH_OK;
}

// ----------------------------------------------------------------------------- orthogonal period powers
int ph_orth_powers(ph_ctx* c, const void* x, int dtype, int64_t W, int N, int max_p, int normalize,
                   unsigned flags, double* autocorr, double* eq3, double* powers) {
  PH_TRY(check_common(c, x, dtype, W, N));
  if (!powers) return fail(PH_E_ARG, "powers is NULL");
  if (max_p < 0) max_p = N / 2;  // QOPeriods.py:1204-1207
  if (max_p < 2) return fail(PH_E_ARG, "max_p=%d must be >= 2", max_p);
  PH_HIP(hipSetDevice(c->device));
  const size_t sz = elem_size(dtype);
  // window + autocorrelation + clipped eq. 3 values in LDS when they fit, otherwise the window is read
  // from HBM / L2 and the two work arrays live in an HBM workspace
  size_t lds = carve_bytes(N, sz) + carve_bytes(N, 8) + carve_bytes(max_p, 8);
  const bool lds_window = lds <= (size_t)c->lds_limit;
  double* gws = nullptr;
  if (!lds_window) {
    lds = 0;
    PH_TRY(ensure(c, c->buf[B_WS1], (size_t)W * ((size_t)N + max_p) * sizeof(double)));
    gws = static_cast<double*>(c->buf[B_WS1].p);
  }
  // divisors d of q with mu(q/d) != 0, for q < max_p
  std::vector<int32_t> mu(max_p, 1), off(max_p + 1, 0), dd, dm;
  {
    std::vector<char> comp(max_p, 0);
    for (int i = 2; i < max_p; ++i) {
      if (comp[i]) continue;
      for (int j = i; j < max_p; j += i) {
        comp[j] = j > i;
        mu[j] = -mu[j];
      }
      for (int64_t j = (int64_t)i * i; j < max_p; j += (int64_t)i * i) mu[j] = 0;
    }
    for (int q = 0; q < max_p; ++q) {
      off[q] = (int32_t)dd.size();
      for (int d = 1; q > 0 && d <= q; ++d)
        if (q % d == 0 && mu[q / d] != 0) {
          dd.push_back(d);
          dm.push_back(mu[q / d]);
        }
    }
    off[max_p] = (int32_t)dd.size();
  }
  const int *d_off, *d_d, *d_mu;
  PH_TRY(upload_table(c, T_AUX0, off.data(), off.size(), &d_off));
  PH_TRY(upload_table(c, T_AUX1, dd.data(), dd.size(), &d_d));
  PH_TRY(upload_table(c, T_AUX2, dm.data(), dm.size(), &d_mu));
  Stage st(c, flags);
  const void* dx;
  void *dr, *de, *dp;
  PH_TRY(st.in(x, (size_t)W * N * sz, &dx));
  PH_TRY(st.out(B_OUT0, autocorr, (size_t)W * N * sizeof(double), &dr));
  PH_TRY(st.out(B_OUT1, eq3, (size_t)W * max_p * sizeof(double), &de));
  PH_TRY(st.out(B_OUT2, powers, (size_t)W * max_p * sizeof(double), &dp));
  const dim3 grid((unsigned)W);
  PH_TRY(dispatch(dtype, lds_window, [&](auto t, auto lw) {
    using T = decltype(t);
    auto kernel = ph::k_orth_powers<T, decltype(lw)::value>;
    PH_TRY(allow_lds(kernel, lds));
    ProfScope ps_(c, "k_orth_powers");
    hipLaunchKernelGGL(kernel, grid, dim3(kBlockWide), lds, c->stream, (const T*)dx, N, max_p, normalize, d_off, d_d, d_mu,
                       gws, (double*)dr, (double*)de, (double*)dp);
    return (int)PH_OK;
  }));
  PH_TRY(launch_check("k_orth_powers"));
  return st.finish();
}

}  // extern "C"
