/*
 * d2q9_oracle_cli.c — command-line front end of the CPU restatement (TEST INFRASTRUCTURE ONLY).
 *
 * Same contract as the reference's main (d2q9-bgk.c:153-440): two positional arguments, writes
 * final_state.dat and av_vels.dat into the cwd, prints the five stdout lines of :411-415.
 * Used to (1) prove byte-identity with oracle/_ref/d2q9-bgk_ref (tests/golden/make_fixtures.py) and
 * (2) time the CPU baseline.  Environment knobs (all optional, none changes results except
 * ORACLE_FAST with >1 thread, which changes only av_vels' float summation order):
 *   ORACLE_THREADS=n   row-parallel OpenMP threads (default 1 = the reference's serial loop)
 *   ORACLE_FAST=1      per-thread float accumulators (the timed CPU-baseline form)
 *   ORACLE_STEPS=n     run n steps instead of the deck's maxIters
 *   ORACLE_NO_OUTPUT=1 skip the two output files (like the reference's -DPROFILE, :419-421)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include "d2q9_oracle.h"

static void die(const char* message, int line, const char* file)   /* d2q9-bgk.c:1145-1151 */
{
  fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  fprintf(stderr, "%s\n", message);
  fflush(stderr);
  exit(EXIT_FAILURE);
}

static double wall(void)
{
  struct timeval t;
  gettimeofday(&t, NULL);
  return t.tv_sec + t.tv_usec / 1000000.0;
}

static int env_int(const char* name, int dflt)
{
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

int main(int argc, char** argv)
{
  if (argc != 3) {                                                       /* :197-200, :1153-1157 */
    fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", argv[0]);
    return EXIT_FAILURE;
  }
  char err[1200];
  oracle_params p;
  if (oracle_read_params(argv[1], &p, err, sizeof err)) die(err, __LINE__, __FILE__);
  int* obstacles = (int*)malloc(sizeof(int) * (size_t)p.nx * p.ny);
  if (!obstacles) die("cannot allocate column memory for obstacles", __LINE__, __FILE__);
  int free_cells = 0;
  if (oracle_read_obstacles(argv[2], p.nx, p.ny, obstacles, &free_cells, err, sizeof err))
    die(err, __LINE__, __FILE__);

  const int threads = env_int("ORACLE_THREADS", 1);
  const int steps = env_int("ORACLE_STEPS", p.max_iters);
  const int fast = env_int("ORACLE_FAST", 0);
  float* cells = (float*)malloc(sizeof(float) * (size_t)p.nx * p.ny * ORACLE_NSPEEDS);
  float* av_vels = (float*)calloc((size_t)steps + 1, sizeof(float));
  if (!cells || !av_vels) die("cannot allocate memory for cells", __LINE__, __FILE__);

  const double tic = wall();                                              /* :278-279 */
  const int rc = fast ? oracle_run_fast(&p, obstacles, free_cells, steps, threads, cells, av_vels)
                      : oracle_run(&p, obstacles, free_cells, steps, threads, cells, av_vels, NULL);
  const double toc = wall();                                              /* :397-398 */
  if (rc) die("cannot allocate memory for cells", __LINE__, __FILE__);

  struct rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  const double usr = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1000000.0;
  const double sys = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1000000.0;

  const float free_cells_inv = 1.0f / free_cells;
  const float av = oracle_av_velocity_sum(&p, cells, obstacles, p.ny) * free_cells_inv;   /* :753 */
  printf("==done==\n");                                                   /* :411-415 */
  printf("Reynolds number:\t\t%.12E\n", oracle_reynolds(&p, av));
  printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usr);
  printf("Elapsed system CPU time:\t%.6lf (s)\n", sys);
  printf("MLUPS:\t\t\t\t%.3f (threads=%d)\n", (double)p.nx * p.ny * steps / (toc - tic) / 1e6, threads);

  if (!env_int("ORACLE_NO_OUTPUT", 0)) {
    if (oracle_write_final_state("final_state.dat", &p, cells, obstacles, p.ny, 0, 0))
      die("could not open file output file", __LINE__, __FILE__);        /* :1061 */
    if (oracle_write_av_vels("av_vels.dat", av_vels, steps))
      die("could not open file output file", __LINE__, __FILE__);        /* :1131 */
  }
  free(cells);
  free(av_vels);
  free(obstacles);
  return EXIT_SUCCESS;
}
