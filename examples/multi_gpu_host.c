/* multi_gpu_host.c -- a plain C host driving libhector_amd.so over every GPU of a node: the
 * call sequence an R (Rcpp) or C++ embedding of the reference would add next to its
 * newcore / setvar / run / fetchvars loop (src/rcpp_hector.cpp:31-356).
 *
 *   cc -std=c99 -Iinclude examples/multi_gpu_host.c -Lhector_amd/lib -lhector_amd \
 *      -Wl,-rpath,$PWD/hector_amd/lib -lm -o multi_gpu_host
 *   ./multi_gpu_host hector_amd/data/ssp245.hxs 1048576 0,1,2,3,4,5,6,7
 *
 * One handle over the device list (hx_newcore_devices): members in contiguous blocks, one per
 * GPU; hx_run queues all of them; hx_ensemble_stats reduces every block on its GPU and combines
 * the blocks with one RCCL all-gather issued by the library.  No Python, no torch. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hector_amd.h"

static void ck(int rc) {
  if (rc) { fprintf(stderr, "hector_amd: %s\n", hx_last_error()); exit(2); }
}

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <scenario> <members> <dev,dev,...> [run-to]\n", argv[0]); return 1; }
  const int n = atoi(argv[2]);
  int devices[64], ndev = 0;
  for (char *tok = strtok(argv[3], ","); tok && ndev < 64; tok = strtok(NULL, ",")) devices[ndev++] = atoi(tok);
  const int run_to = argc > 4 ? atoi(argv[4]) : 2300;

  hx_core *core = NULL;
  ck(hx_newcore_devices(argv[1], n, devices, ndev, &core));
  int shards = 0, dev[64], off[65];
  ck(hx_shards(core, &shards, dev, off));
  for (int s = 0; s < shards; ++s)
    printf("shard %d: GPU %d, members %d..%d\n", s, dev[s], off[s], off[s + 1] - 1);

  /* a perturbed-parameter ensemble: climate sensitivity 1.5 .. 6 K, Q10 1 .. 3 */
  double *S = malloc(sizeof(double) * n), *q10 = malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) {
    S[i] = 1.5 + 4.5 * (i + 0.5) / n;
    q10[i] = 1.0 + 2.0 * fmod(i * 0.6180339887498949, 1.0);
  }
  ck(hx_setvar(core, "S", S, n, "degC"));
  ck(hx_setvar(core, "q10_rh", q10, n, "(unitless)"));
  const char *outs[2] = {"CO2_concentration", "global_tas"};
  ck(hx_set_outputs(core, 2, outs));

  ck(hx_run(core, (double)run_to));   /* every GPU's kernels are queued when this returns */
  ck(hx_sync(core));
  double ms = 0;
  ck(hx_last_run_ms(core, &ms));

  int start = 0;
  ck(hx_dates(core, &start, NULL, NULL));
  const int ny = run_to - start + 1;
  double *st = malloc(sizeof(double) * 2 * ny * 5);
  ck(hx_ensemble_stats(core, 2, outs, start, run_to, st, NULL));   /* the one collective */
  int world = 0; const char *backend = "";
  ck(hx_comm_info(core, &world, NULL, &backend));
  for (int v = 0; v < 2; ++v) {
    const double *row = st + ((size_t)v * ny + (ny - 1)) * 5;   /* last year */
    const double mean = row[1] / row[0], var = row[2] / row[0] - mean * mean;
    printf("%s %d: members %.0f mean %.6f sd %.6f min %.6f max %.6f\n", outs[v], run_to, row[0], mean,
           sqrt(var > 0 ? var : 0), row[3], row[4]);
  }
  printf("kernel %.3f ms on the slowest GPU; statistics over %d rank(s), %s\n", ms, world > 0 ? world : 1,
         world > 0 ? backend : "no collective");

  /* a few members' trajectories, in member order whatever GPU they ran on */
  double *tas = malloc(sizeof(double) * n);
  ck(hx_fetchvars(core, "global_tas", run_to, run_to, tas));
  printf("global_tas(%d) of members 0, %d, %d: %.6f %.6f %.6f\n", run_to, n / 2, n - 1, tas[0], tas[n / 2], tas[n - 1]);
  unsigned *status = malloc(sizeof(unsigned) * n);
  ck(hx_status(core, status));
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += status[i] != 0;
  printf("members with model errors: %d\n", bad);
  ck(hx_shutdown(core));
  free(S); free(q10); free(st); free(tas); free(status);
  return bad != 0;
}
