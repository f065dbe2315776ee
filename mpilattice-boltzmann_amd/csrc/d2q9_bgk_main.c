/* d2q9_bgk_main.c — the thin C host shim: the reference's command-line contract (d2q9-bgk.c:153-440) in front of
 * liblbm_d2q9.so.  Plain C99 (gcc -std=c99), nothing but the C ABI of include/lbm_d2q9.h + lbm_d2q9_p2p.h.
 *
 *   d2q9-bgk <paramfile> <obstaclefile>
 *
 * writes final_state.dat and av_vels.dat into the cwd (d2q9-bgk.c:63-64) and prints the five stdout lines of
 * d2q9-bgk.c:411-415 byte-compatibly, followed by extra lines (MLUPS, roofline).  Everything between tic and toc is
 * the device path; this file only parses, times and writes.
 * Environment (optional): LBM_DEVICE=<hip ordinal>, LBM_NO_OUTPUT=1 (like the reference's -DPROFILE build, :419-421),
 * LBM_FLAGS=<lbm_create flags> (default LBM_FLAG_EXACT_AVVELS: the contract path forms every sum|u| term as the reference does,
 * sqrt((double)u_sq) * densinv in double precision, d2q9-bgk.c:667; LBM_FLAGS=0 selects the library's default, compensated float
 * sums of relative error ~2^-44 per term, 1 - 2.5 % faster on the large decks, the same av_vels floats on every deck tested).
 *
 * LBM_GPUS=N (N > 1) plays the role of `mpirun -np N` (mpi_submit:63) inside ONE process: the rows are partitioned by
 * the reference's rule (d2q9-bgk.c:834-862), rank r lives on device LBM_DEVICES[r] (a comma list; default r), one host
 * thread per rank drives its device, and the halos travel as direct peer-to-peer stores over xGMI
 * (include/lbm_d2q9_p2p.h).  The ranks' observables are gathered in rank order — the order in which the reference's
 * ranks append to final_state.dat (:1049-1057).
 * LBM_RANK_GRID=PXxPY (PX * PY = N; or `auto`: lbm_choose_rank_grid decides between rows and tiles) runs those N ranks as the tile (2-D) decomposition instead (lbm_create_tile): rank = ry * PX + rx owns a
 * block of columns of a block of rows; the reference's report discusses such a split for grids wider than tall and never built it. */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include "lbm_d2q9.h"
#include "lbm_d2q9_p2p.h"

static void die(const char* message, const int line, const char* file)   /* d2q9-bgk.c:1145-1151 */
{
  fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  fprintf(stderr, "%s\n", message);
  fflush(stderr);
  exit(EXIT_FAILURE);
}

static void usage(const char* exe)                                        /* d2q9-bgk.c:1153-1157 */
{
  fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", exe);
  exit(EXIT_FAILURE);
}

static double wall_seconds(void)
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

static void* xmalloc(size_t bytes)
{
  void* p = malloc(bytes ? bytes : 1);
  if (!p) die("cannot allocate host memory", __LINE__, __FILE__);
  return p;
}

/* one rank of a single-process multi-GPU run: what an MPI rank of the reference does between tic and toc */
typedef struct rank_job {
  lbm_p2p* ring;
  int n_steps;
  double* tot;
  int failed;
  char message[512];
} rank_job;

static void* rank_main(void* arg)
{
  rank_job* job = (rank_job*)arg;
  if (lbm_p2p_run(job->ring, job->n_steps, job->tot)) {                   /* :315-396 */
    job->failed = 1;
    strncpy(job->message, lbm_last_error(), sizeof job->message - 1);     /* the error text is per thread */
  }
  return NULL;
}

int main(int argc, char* argv[])
{
  lbm_params params;
  int* obstacles;
  int free_cells = 0;
  float* av_vels;
  float* obs;                    /* u_x, u_y, u, pressure per cell of the whole grid */
  double tic = 0.0, toc = 0.0, usrtim, systim, mlups;
  struct rusage ru;
  float free_cells_inv, av;
  int ngpus, tiled = 0;
  unsigned flags;
  size_t nx;

  if (argc != 3) usage(argv[0]);                                          /* :197-200 */

  if (lbm_read_params(argv[1], &params)) die(lbm_last_error(), __LINE__, __FILE__);
  if (params.nx <= 0 || params.ny <= 0 || params.max_iters < 0) die("could not read param file: nx", __LINE__, __FILE__);
  nx = (size_t)params.nx;
  obstacles = (int*)xmalloc(sizeof(int) * nx * (size_t)params.ny);
  if (lbm_read_obstacles(argv[2], params.nx, params.ny, obstacles, &free_cells)) die(lbm_last_error(), __LINE__, __FILE__);

  ngpus = env_int("LBM_GPUS", 1);
  flags = (unsigned)env_int("LBM_FLAGS", (int)LBM_FLAG_EXACT_AVVELS);
  av_vels = (float*)xmalloc(sizeof(float) * ((size_t)params.max_iters + 1));
  obs = (float*)xmalloc(sizeof(float) * nx * (size_t)params.ny * 4);

  if (ngpus <= 1) {
    lbm_ctx* ctx = NULL;
    if (lbm_create(&ctx, &params, free_cells, obstacles, 0, params.ny, env_int("LBM_DEVICE", 0), flags))
      die(lbm_last_error(), __LINE__, __FILE__);
    tic = wall_seconds();                                                  /* :278-279 */
    if (lbm_run(ctx, params.max_iters, av_vels)) die(lbm_last_error(), __LINE__, __FILE__);
    toc = wall_seconds();                                                  /* :397-398 */
    if (lbm_get_observables(ctx, obs)) die(lbm_last_error(), __LINE__, __FILE__);
    lbm_destroy(ctx);
  } else {
    /* ---- one process, N GPUs: N ranks of the reference's decomposition, one host thread each ------------------- */
    int* device;
    lbm_layout* lay;
    lbm_ctx** ctx;
    rank_job* job;
    pthread_t* thread;
    char* handles;
    const char* list = getenv("LBM_DEVICES");
    const char* grid = getenv("LBM_RANK_GRID");
    lbm_tile_layout* tile = NULL;
    int r, q, t, most = 1, px = 0, py = 0;
    if (grid && strcmp(grid, "auto") == 0) {                              /* the library's choice between row blocks and tiles */
      if (lbm_choose_rank_grid(&params, ngpus, flags, &px, &py)) die(lbm_last_error(), __LINE__, __FILE__);
      if (px == 1) grid = NULL;
    }
    if (grid && *grid) {
      if (strcmp(grid, "auto") != 0 && (sscanf(grid, "%dx%d", &px, &py) != 2 || px < 1 || py < 1 || px * py != ngpus))
        die("LBM_RANK_GRID: expected PXxPY with PX * PY = LBM_GPUS, or auto", __LINE__, __FILE__);
      tile = (lbm_tile_layout*)xmalloc(sizeof(lbm_tile_layout) * (size_t)ngpus);
    }
    if (ngpus > 64) die("LBM_GPUS: at most 64 ranks (MPI_PROCS, d2q9-bgk.c:67)", __LINE__, __FILE__);
    device = (int*)xmalloc(sizeof(int) * (size_t)ngpus);
    lay = (lbm_layout*)xmalloc(sizeof(lbm_layout) * (size_t)ngpus);
    ctx = (lbm_ctx**)xmalloc(sizeof(lbm_ctx*) * (size_t)ngpus);
    job = (rank_job*)xmalloc(sizeof(rank_job) * (size_t)ngpus);
    thread = (pthread_t*)xmalloc(sizeof(pthread_t) * (size_t)ngpus);
    handles = (char*)xmalloc((size_t)ngpus * LBM_P2P_HANDLE_BYTES);
    for (r = 0; r < ngpus; ++r) {
      device[r] = r;
      if (list && *list) {
        device[r] = atoi(list);
        list = strchr(list, ',');
        if (list) ++list;
      }
    }
    /* Several ranks on ONE device (a test set-up; a node run has one rank per GPU): each rank's stream needs a
     * hardware queue of its own, or a rank's wait kernel can end up queued in front of the push it waits for.  The HIP
     * runtime reads its queue budget from the environment when it initialises, which is still ahead. */
    for (r = 0; r < ngpus; ++r) {
      int same = 0;
      for (q = 0; q < ngpus; ++q) same += device[q] == device[r];
      if (same > most) most = same;
    }
    if (most > 1) {
      char budget[16];
      snprintf(budget, sizeof budget, "%d", 2 * most + 4);
      setenv("GPU_MAX_HW_QUEUES", budget, 0);
    }
    tiled = tile != NULL;
    for (r = 0; r < ngpus && tile; ++r) {
      /* the block this rank needs: its rows and columns plus the ghost rows / columns around them, wrapping both ways */
      int rows, cols, i, j;
      int* window;
      if (lbm_tile_layout_of(&params, px, py, r, flags, &tile[r])) die(lbm_last_error(), __LINE__, __FILE__);
      rows = tile[r].ny_local + 2 * tile[r].ghost_y;
      cols = tile[r].nx_local + 2 * tile[r].ghost_x;
      window = (int*)xmalloc(sizeof(int) * (size_t)rows * (size_t)cols);
      for (i = 0; i < rows; ++i) {
        int g = (tile[r].y0 - tile[r].ghost_y + i) % params.ny;
        if (g < 0) g += params.ny;
        for (j = 0; j < cols; ++j) {
          int x = (tile[r].x0 - tile[r].ghost_x + j) % params.nx;
          if (x < 0) x += params.nx;
          window[(size_t)i * (size_t)cols + (size_t)j] = obstacles[(size_t)g * nx + (size_t)x];
        }
      }
      if (lbm_create_tile(&ctx[r], &params, free_cells, window, px, py, r, device[r], flags)) die(lbm_last_error(), __LINE__, __FILE__);
      free(window);
    }
    for (r = 0; r < ngpus; ++r) {
      int rows, i;
      int* window;
      if (tile) {
        memset(&job[r], 0, sizeof job[r]);
        if (lbm_p2p_create(&job[r].ring, ctx[r], ngpus, r)) die(lbm_last_error(), __LINE__, __FILE__);
        if (lbm_p2p_handle(job[r].ring, handles + (size_t)r * LBM_P2P_HANDLE_BYTES)) die(lbm_last_error(), __LINE__, __FILE__);
        job[r].n_steps = params.max_iters;
        job[r].tot = (double*)xmalloc(sizeof(double) * ((size_t)params.max_iters + 1));
        continue;
      }
      if (lbm_rank_layout(&params, ngpus, r, flags, &lay[r])) die(lbm_last_error(), __LINE__, __FILE__);
      /* the rows this rank needs: its own plus `ghost` rows below and above, wrapping (the scatter of :968-970) */
      rows = lay[r].ny_local + 2 * lay[r].ghost;
      window = (int*)xmalloc(sizeof(int) * (size_t)rows * nx);
      for (i = 0; i < rows; ++i) {
        int g = (lay[r].y0 - lay[r].ghost + i) % params.ny;
        if (g < 0) g += params.ny;
        memcpy(window + (size_t)i * nx, obstacles + (size_t)g * nx, sizeof(int) * nx);
      }
      if (lbm_create_rank(&ctx[r], &params, free_cells, window, ngpus, r, device[r], flags)) die(lbm_last_error(), __LINE__, __FILE__);
      free(window);
      memset(&job[r], 0, sizeof job[r]);
      if (lbm_p2p_create(&job[r].ring, ctx[r], ngpus, r)) die(lbm_last_error(), __LINE__, __FILE__);
      if (lbm_p2p_handle(job[r].ring, handles + (size_t)r * LBM_P2P_HANDLE_BYTES)) die(lbm_last_error(), __LINE__, __FILE__);
      job[r].n_steps = params.max_iters;
      job[r].tot = (double*)xmalloc(sizeof(double) * ((size_t)params.max_iters + 1));
    }
    for (r = 0; r < ngpus; ++r)
      if (lbm_p2p_connect(job[r].ring, handles)) die(lbm_last_error(), __LINE__, __FILE__);
    tic = wall_seconds();                                                  /* :278-279 */
    for (r = 0; r < ngpus; ++r)
      if (pthread_create(&thread[r], NULL, rank_main, &job[r])) die("cannot start a rank thread", __LINE__, __FILE__);
    for (r = 0; r < ngpus; ++r) pthread_join(thread[r], NULL);
    toc = wall_seconds();                                                  /* :397-398 */
    for (r = 0; r < ngpus; ++r)
      if (job[r].failed) die(job[r].message, __LINE__, __FILE__);
    {
      const float inv = 1.0f / free_cells;                                 /* :950 */
      for (t = 0; t < params.max_iters; ++t) av_vels[t] = (float)(job[0].tot[t] * (double)inv);   /* :367 */
    }
    for (r = 0; r < ngpus; ++r) {
      if (tile) {                                                          /* the rank's block, row by row, into its place */
        const size_t w = (size_t)tile[r].nx_local;
        float* block = (float*)xmalloc(sizeof(float) * 4 * w * (size_t)tile[r].ny_local);
        int i;
        if (lbm_get_observables(ctx[r], block)) die(lbm_last_error(), __LINE__, __FILE__);
        for (i = 0; i < tile[r].ny_local; ++i)
          memcpy(obs + ((size_t)(tile[r].y0 + i) * nx + (size_t)tile[r].x0) * 4, block + (size_t)i * w * 4, sizeof(float) * 4 * w);
        free(block);
      } else if (lbm_get_observables(ctx[r], obs + (size_t)lay[r].y0 * nx * 4)) die(lbm_last_error(), __LINE__, __FILE__);
      lbm_p2p_destroy(job[r].ring);
      free(job[r].tot);
    }
    for (r = 0; r < ngpus; ++r) lbm_destroy(ctx[r]);
    free(device); free(lay); free(ctx); free(job); free(thread); free(handles); free(tile);
  }
  getrusage(RUSAGE_SELF, &ru);                                             /* :399-403 */
  usrtim = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1000000.0;
  systim = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1000000.0;

  free_cells_inv = 1.0f / free_cells;                                      /* :950 */
  av = lbm_av_velocity_obs(&params, obs, obstacles, params.ny) * free_cells_inv;   /* :753 */
  printf("==done==\n");                                                    /* :411-415 */
  printf("Reynolds number:\t\t%.12E\n", lbm_reynolds(&params, av));
  printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usrtim);
  printf("Elapsed system CPU time:\t%.6lf (s)\n", systim);
  mlups = (double)params.nx * params.ny * params.max_iters / (toc - tic) / 1e6;
  printf("MLUPS:\t\t\t\t%.1f (%d GPU%s%s)\n", mlups, ngpus > 1 ? ngpus : 1, ngpus > 1 ? "s, peer-to-peer halos" : "",
         tiled ? ", tile decomposition" : "");
  printf("HBM roofline (108 B/cell-step @ 8.0 TB/s = 74074 MLUPS per GPU):\t%.1f %%\n",
         100.0 * mlups / (ngpus > 1 ? ngpus : 1) / (8.0e12 / 108.0 / 1e6));

  if (!env_int("LBM_NO_OUTPUT", 0)) {                                      /* :419-421 */
    if (lbm_write_final_state_obs("final_state.dat", &params, obs, obstacles, params.ny, 0, 0))
      die(lbm_last_error(), __LINE__, __FILE__);
    if (lbm_write_av_vels("av_vels.dat", av_vels, params.max_iters)) die(lbm_last_error(), __LINE__, __FILE__);
  }
  free(obs); free(av_vels); free(obstacles);
  return EXIT_SUCCESS;                                                     /* :439 */
}
