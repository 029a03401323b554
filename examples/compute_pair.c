/* compute_pair.c -- the whole ICP::compute of laser_slam (laser_slam/src/laser_track.cpp:496) through the C ABI,
 * from plain C.  Usage: compute_pair <reading.bin> <reference.bin>   (float32 x,y,z,1 per point)
 * Build: gcc -std=c99 -I include examples/compute_pair.c -L laser_slam_amd -llsgpu_icp -Wl,-rpath,$PWD/laser_slam_amd */
#include <stdio.h>
#include <stdlib.h>

#include "lsgpu_icp.h"

static float* read_cloud(const char* path, int64_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  float* p = (float*)malloc((size_t)bytes);
  if (fread(p, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); exit(2); }
  fclose(f);
  *n = bytes / 16;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s reading.bin reference.bin\n", argv[0]); return 2; }
  int64_t nq = 0, nr = 0;
  float* reading = read_cloud(argv[1], &nq);
  float* reference = read_cloud(argv[2], &nr);

  lsgpu_icp_config cfg;
  lsgpu_icp_config_yaml(&cfg);                 /* the chain of laser_slam/configurations/icp_default.yaml */
  lsgpu_chain_config chain;
  lsgpu_chain_config_yaml(&chain);
  chain.seed = 0;                              /* reproducible sampling */

  lsgpu_icp* h = NULL;
  int rc = lsgpu_icp_create(&cfg, /*device*/ 0, &h);
  if (rc != LSGPU_OK) { fprintf(stderr, "lsgpu_icp_create: %s (no ROCm GPU?)\n", lsgpu_strerror(rc)); return 1; }

  const float T_init[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   /* column major */
  float T[16];
  lsgpu_icp_stats st;
  rc = lsgpu_icp_compute(h, reading, nq, reference, nr, T_init, &chain, T, &st);
  if (rc == LSGPU_NO_CONVERGENCE) fprintf(stderr, "no convergence (%s): T = T_init\n", lsgpu_last_error(h));
  else if (rc != LSGPU_OK) { fprintf(stderr, "compute: %s (%s)\n", lsgpu_strerror(rc), lsgpu_last_error(h)); return 1; }

  printf("iterations %d converged %d filters+grid %.2f ms loop %.2f ms\n", st.iterations, st.converged, st.t_reserved[0],
         st.t_total_ms);
  for (int r = 0; r < 4; ++r) printf("%12.6f %12.6f %12.6f %12.6f\n", T[r], T[4 + r], T[8 + r], T[12 + r]);
  lsgpu_icp_destroy(h);
  free(reading);
  free(reference);
  return rc == LSGPU_OK ? 0 : 3;
}
