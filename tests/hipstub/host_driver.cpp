// Drives every entry point of include/periodhip.h through the HOST half of the library (hip_stub.cpp stands in for
// the runtime; kernels do not run, outputs are not looked at).  Built with -fsanitize=address,undefined by
// tests/test_host_sanitizers.py: what is checked is that argument validation, pass plans, geometry / CSR / Bluestein
// tables, staging copies and LDS layout arithmetic touch no byte out of bounds and hit no undefined behaviour.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/periodhip.h"

static int fails = 0;
#define EXPECT(call, want)                                                              \
  do {                                                                                  \
    const int rc_ = (call);                                                             \
    if (rc_ != (want)) {                                                                \
      std::printf("FAIL %s:%d %s -> %d (%s), want %d\n", __FILE__, __LINE__, #call, rc_, ph_last_error(), (want)); \
      ++fails;                                                                          \
    }                                                                                   \
  } while (0)

struct Csr {
  std::vector<int32_t> off, q;
};

// proper divisors d of p with 1 < d < p (any fixed order will do here)
static Csr factor_tables(int max_p) {
  Csr t;
  t.off.assign(max_p + 2, 0);
  for (int p = 0; p <= max_p; ++p) {
    t.off[p] = (int32_t)t.q.size();
    for (int d = 2; p > 0 && d < p; ++d)
      if (p % d == 0) t.q.push_back(d);
  }
  t.off[max_p + 1] = (int32_t)t.q.size();
  if (t.q.empty()) t.q.push_back(1);
  return t;
}

// sub-periods p / f for the prime proper divisors f of p (Periods.py:209-214)
static Csr orth_tables(int max_p) {
  Csr t;
  t.off.assign(max_p + 2, 0);
  for (int p = 0; p <= max_p; ++p) {
    t.off[p] = (int32_t)t.q.size();
    for (int f = 2; p > 0 && f < p; ++f) {
      bool prime = true;
      for (int d = 2; d * d <= f; ++d) prime &= (f % d != 0);
      if (prime && p % f == 0) t.q.push_back(p / f);
    }
  }
  t.off[max_p + 1] = (int32_t)t.q.size();
  if (t.q.empty()) t.q.push_back(1);
  return t;
}

int main() {
  ph_ctx* c = nullptr;
  int n = 0;
  EXPECT(ph_device_count(&n), PH_OK);
  EXPECT(ph_create(0, &c), PH_OK);
  EXPECT(ph_create(0, nullptr), PH_E_ARG);
  int cu = 0, lds = 0, a = 0, b = 0;
  EXPECT(ph_device_info(c, &cu, &lds), PH_OK);
  EXPECT(ph_max_window(c, PH_F64, 0, &a), PH_OK);
  EXPECT(ph_max_window(c, PH_F32, PH_FLAG_TRUNC | PH_FLAG_ORTH, &a), PH_OK);
  for (int lo : {1, 2, 63, 64, 65, 700}) {
    for (int hi : {lo, lo + 1, 3 * lo + 77, 5000}) EXPECT(ph_sweep_plan_info(c, lo, hi, &a, &b), PH_OK);
  }
  EXPECT(ph_sweep_plan_info(c, 0, 5, &a, &b), PH_E_ARG);
  EXPECT(ph_m_best_info(c, PH_F64, 4096, 10, 2, -1, 0, &a, &b), PH_OK);
  EXPECT(ph_m_best_info(c, PH_F32, 4096, 10, 2, 1365, PH_FLAG_TRUNC, &a, &b), PH_OK);
  EXPECT(ph_m_best_plan_info(c, PH_F64, 4096, 10, 2, -1, 0, &a, &b), PH_OK);
  EXPECT(ph_m_best_plan_info(c, PH_F64, 4096, 10, 9, 3, 0, &a, &b), PH_E_ARG);
  EXPECT(ph_profile_enable(c, 1), PH_OK);

  const int sizes[][2] = {{1, 7}, {3, 100}, {5, 1000}, {2, 4096}, {1, 9000}, {1, 20000}};
  for (const auto& sz : sizes) {
    const int W = sz[0], N = sz[1];
    std::vector<double> x((size_t)W * N);
    for (size_t i = 0; i < x.size(); ++i) x[i] = std::sin(0.37 * (double)i) + 0.01 * (double)(i % 7);
    std::vector<float> xf(x.begin(), x.end());
    const int maxp = N;  // tables up to N
    const Csr fac = factor_tables(maxp), orth = orth_tables(maxp);
    for (int dtype : {PH_F64, PH_F32}) {
      const void* px = dtype == PH_F64 ? (const void*)x.data() : (const void*)xf.data();
      const size_t es = dtype == PH_F64 ? 8 : 4;
      for (unsigned dev : {0u, (unsigned)PH_FLAG_DEVICE}) {
        std::vector<double> out((size_t)W * 8 * N + 16);
        std::vector<int32_t> pl = {1, 2, 3, N / 2 > 0 ? N / 2 : 1, N, N + 3, 64, 65};
        EXPECT(ph_periodic_norm(c, px, dtype, W, N, 0, dev, out.data()), PH_OK);
        EXPECT(ph_periodic_norm(c, px, dtype, W, N, 5, dev, out.data()), PH_OK);
        for (unsigned fl : {0u, (unsigned)PH_FLAG_TRUNC, (unsigned)PH_FLAG_ORTH, (unsigned)(PH_FLAG_TRUNC | PH_FLAG_ORTH | PH_FLAG_SINGLE)}) {
          std::vector<int32_t> pl2(pl);
          if (fl & PH_FLAG_ORTH)
            for (auto& p : pl2) p = p > maxp ? maxp : p;
          EXPECT(ph_project_batch(c, px, dtype, W, N, pl2.data(), (int)pl2.size(), orth.off.data(), orth.q.data(), maxp, fl | dev,
                                  out.data()), PH_OK);
        }
        const int p_hi = N / 3 > 2 ? N / 3 : 2;
        std::vector<double> sw((size_t)W * (p_hi + 2));
        for (int mode : {PH_SWEEP_NORM, PH_SWEEP_NORM_GAMMA, PH_SWEEP_MAXABS}) {
          EXPECT(ph_sweep(c, px, dtype, W, N, 2, p_hi, mode, nullptr, nullptr, 0, dev, sw.data()), PH_OK);
          EXPECT(ph_sweep(c, px, dtype, W, N, 1, p_hi, mode, orth.off.data(), orth.q.data(), maxp, dev | PH_FLAG_ORTH, sw.data()),
                 PH_OK);
        }
        const int num = 5;
        std::vector<uint32_t> per((size_t)W * num);
        std::vector<double> pw((size_t)W * num), bases((size_t)W * num * N);
        std::vector<int32_t> st(W), nsw(W), cnt((size_t)2 * W), ips((size_t)W * 64);
        for (int gamma : {0, 1})
          for (unsigned fl : {0u, (unsigned)(PH_FLAG_TRUNC | PH_FLAG_ORTH)}) {
            EXPECT(ph_m_best(c, px, dtype, W, N, num, 2, p_hi, gamma, orth.off.data(), orth.q.data(), fac.off.data(), fac.q.data(),
                             maxp, fl | dev, per.data(), pw.data(), bases.data(), st.data(), nsw.data()), PH_OK);
          }
        EXPECT(ph_m_best(c, px, dtype, W, N, num, 2, -1, 0, nullptr, nullptr, fac.off.data(), fac.q.data(), maxp, dev, per.data(),
                         pw.data(), bases.data(), st.data(), nullptr), PH_OK);
        EXPECT(ph_m_best(c, px, dtype, W, N, num, 2, p_hi, 0, nullptr, nullptr, nullptr, nullptr, maxp, dev, per.data(), pw.data(),
                         bases.data(), st.data(), nullptr), PH_E_ARG);
        std::vector<double> spw((size_t)W * 16), sb((size_t)W * 16 * N);
        EXPECT(ph_small_to_large(c, px, dtype, W, N, 0.05, -1, nullptr, nullptr, 0, dev, 16, cnt.data(), ips.data(), spw.data(),
                                 sb.data(), st.data()), PH_OK);
        EXPECT(ph_small_to_large(c, px, dtype, W, N, 0.05, p_hi, orth.off.data(), orth.q.data(), maxp, dev | PH_FLAG_ORTH | PH_FLAG_NOSYNC,
                                 16, cnt.data(), ips.data(), spw.data(), nullptr, st.data()), PH_OK);
        EXPECT(ph_best_correlation(c, px, dtype, W, N, num, -1, 0.01, nullptr, nullptr, 0, dev, per.data(), pw.data(), bases.data(),
                                   st.data()), N >= 9 ? PH_OK : PH_OK);
        for (int win : {N, 64, 100, 4097})
          EXPECT(ph_best_frequency(c, px, dtype, W, N, win, num, orth.off.data(), orth.q.data(), maxp, dev, per.data(), pw.data(),
                                   bases.data(), st.data()), 2 * win <= maxp || true ? PH_OK : PH_OK);
        if (N >= 16) {
          const int qh = N / 3 < 700 ? N / 3 : 700;
          std::vector<double> rn((size_t)W * (qh + 1));
          EXPECT(ph_ramanujan_norms(c, px, dtype, W, N, 2, qh, dev, rn.data()), PH_OK);
          EXPECT(ph_ramanujan_norms(c, px, dtype, W, N, 5, 9, dev, rn.data()), PH_OK);
        }
        std::vector<int32_t> fp = {3, 8, N / 4 > 0 ? N / 4 : 1}, fk = {3, 7, N / 4 > 1 ? N / 4 - 1 : 1};
        const int rows = fk[0] + fk[1] + fk[2];
        std::vector<double> fs((size_t)W * rows), ts((size_t)W * N);
        EXPECT(ph_fold_sums(c, px, dtype, W, N, fp.data(), fk.data(), 3, dev, fs.data()), PH_OK);
        EXPECT(ph_tile_sum(c, fs.data(), W, N, fp.data(), fk.data(), 3, dtype, dev, ts.data()), PH_OK);
        for (int kcap : {64, 512, 1024}) {
          int ok = 0;
          EXPECT(ph_qo_feasible(c, dtype, N, p_hi, kcap, &ok), PH_OK);
          if (!ok) continue;
          std::vector<double> w((size_t)W * kcap), nr((size_t)W * num);
          std::vector<int32_t> kp((size_t)W * num);
          std::vector<char> resid((size_t)W * N * es);
          EXPECT(ph_qo_find_periods(c, px, dtype, W, N, num, 0.1, 2, p_hi, kcap, dev, per.data(), nr.data(), kp.data(), cnt.data(),
                                    w.data(), resid.data(), st.data()), PH_OK);
        }
        const int mp = N / 2 > 2 ? N / 2 : 2;
        std::vector<double> ac((size_t)W * N), e3((size_t)W * mp), op((size_t)W * mp);
        EXPECT(ph_orth_powers(c, px, dtype, W, N, mp, 1, dev, ac.data(), e3.data(), op.data()), PH_OK);
        EXPECT(ph_orth_powers(c, px, dtype, W, N, -1, 0, dev, nullptr, nullptr, op.data()), PH_OK);
      }
    }
    // a dictionary for ph_dict_project
    if (N <= 1000) {
      std::vector<double> basis((size_t)6 * N, 0.5);
      std::vector<float> proj((size_t)6 * N);
      EXPECT(ph_dict_project(c, x.data(), basis.data(), 6, N, 0, proj.data()), PH_OK);
    }
    // malformed tables and arguments must be refused, not read
    Csr bad = orth;
    bad.off[3] = bad.off[2] - 1 < 0 ? 5 : bad.off[2] - 1;
    bad.off[4] = 0;
    std::vector<double> out((size_t)W * N);
    std::vector<int32_t> one = {2};
    EXPECT(ph_project_batch(c, x.data(), PH_F64, W, N, one.data(), 1, bad.off.data(), bad.q.data(), maxp, PH_FLAG_ORTH, out.data()),
           PH_E_ARG);
    EXPECT(ph_project_batch(c, x.data(), PH_F64, W, N, one.data(), 1, orth.off.data(), orth.q.data(), 1, PH_FLAG_ORTH, out.data()),
           PH_E_ARG);
    EXPECT(ph_project_batch(c, x.data(), 7, W, N, one.data(), 1, nullptr, nullptr, 0, 0, out.data()), PH_E_ARG);
    EXPECT(ph_project_batch(c, nullptr, PH_F64, W, N, one.data(), 1, nullptr, nullptr, 0, 0, out.data()), PH_E_ARG);
    EXPECT(ph_sweep(c, x.data(), PH_F64, 0, N, 2, 3, 0, nullptr, nullptr, 0, 0, out.data()), PH_E_ARG);
  }
  float ms[300];
  int cntp = 0;
  EXPECT(ph_profile_read(c, ms, 300, &cntp), PH_OK);
  for (int i = 0; i < cntp && i < 256; ++i) (void)ph_profile_name(c, i);
  EXPECT(ph_sync(c), PH_OK);
  EXPECT(ph_destroy(c), PH_OK);
  if (fails) {
    std::printf("host sanitizer driver: %d unexpected return codes\n", fails);
    return 1;
  }
  std::printf("host sanitizer driver ok\n");
  return 0;
}
