// tools/bench_variants.cpp -- sweep kernel variants / layouts of libmhlik on
// one GPU and print likelihood-kernel time and GB/s.  Tuning aid only.
//   build: g++ -O2 -Iinclude tools/bench_variants.cpp -Lmcmc_amd/lib -lmhlik -Wl,-rpath,'$ORIGIN/../mcmc_amd/lib' -o tools/bench_variants
//   usage: bench_variants [N] [d] [reps] [precisions: 1|2|3] [variants "1,2,3" | all] [layouts "1,2"] [timing 0|1]
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "mhlik.h"

static std::vector<int> parse_list(const char *s, int lo, int hi)
{
  std::vector<int> v;
  if (!s || !strcmp(s, "all")) {
    for (int i = lo; i <= hi; i++) v.push_back(i);
    return v;
  }
  std::string str(s);
  size_t pos = 0;
  while (pos < str.size()) {
    size_t c = str.find(',', pos);
    if (c == std::string::npos) c = str.size();
    v.push_back(atoi(str.substr(pos, c - pos).c_str()));
    pos = c + 1;
  }
  return v;
}

int main(int argc, char **argv)
{
  int64_t N = argc > 1 ? atoll(argv[1]) : 10000000;
  int d = argc > 2 ? atoi(argv[2]) : 64;
  int reps = argc > 3 ? atoi(argv[3]) : 20;
  int precs = argc > 4 ? atoi(argv[4]) : 3;
  std::vector<int> variants = parse_list(argc > 5 ? argv[5] : "all", 1, 12);
  std::vector<int> layouts = parse_list(argc > 6 ? argv[6] : "1,2", 1, 2);
  int timing = argc > 7 ? atoi(argv[7]) : 1;

  std::vector<double> theta(d);
  std::vector<float> thetaf(d);
  for (int j = 0; j < d; j++) {
    theta[j] = 0.01 * ((j * 37) % 11 - 5);
    thetaf[j] = (float)theta[j];
  }
  printf("# N=%lld d=%d reps=%d\n", (long long)N, d, reps);
  printf("# prec layout variant  min_ms  med_ms  GB/s(min)  GB/s(med)  pct8TB(med)  wall_ms/call  L\n");
  for (int prec = 1; prec <= 2; prec++) {
    if (!(precs & prec)) continue;
    for (int layout : layouts)
      for (int variant : variants) {
        mhlik_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.n_rows = N;
        cfg.n_cols = d;
        cfg.precision = prec;
        cfg.device = 0;
        cfg.world = 1;
        cfg.layout = layout;
        cfg.variant = variant;
        cfg.timing = timing;
        mhlik_ctx *ctx = nullptr;
        if (mhlik_create(&cfg, &ctx)) {
          printf("create failed: %s\n", mhlik_last_error(nullptr));
          return 1;
        }
        if (mhlik_fill_synthetic(ctx, 12345)) {
          printf("fill failed: %s\n", mhlik_last_error(ctx));
          return 1;
        }
        std::vector<float> ms;
        double L = 0;
        float Lf = 0;
        struct timespec t0, t1;
        for (int r = 0; r < reps + 3; r++) {
          if (r == 3) clock_gettime(CLOCK_MONOTONIC, &t0);
          int rc = prec == 1 ? mhlik_loglik_f64(ctx, theta.data(), &L) : mhlik_loglik_f32(ctx, thetaf.data(), &Lf);
          if (rc) {
            printf("loglik failed: %s\n", mhlik_last_error(ctx));
            return 1;
          }
          float m;
          int64_t b;
          mhlik_last_kernel_ms(ctx, &m, &b);
          if (r >= 3) ms.push_back(m);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double wall = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / reps;
        std::sort(ms.begin(), ms.end());
        double bytes = (double)N * d * (prec == 1 ? 8 : 4);
        double mn = ms.front(), md = ms[ms.size() / 2];
        if (!timing) mn = md = wall;
        printf("%d %d %2d  %8.4f %8.4f  %8.1f %8.1f  %5.1f%%  %8.4f  %.10g\n", prec, layout, variant, mn, md,
               bytes / mn * 1e-6, bytes / md * 1e-6, bytes / md * 1e-6 / 80.0, wall, prec == 1 ? L : (double)Lf);
        fflush(stdout);
        mhlik_destroy(ctx);
      }
  }
  return 0;
}
