// d2q9_bgk_main.cpp — the thin host shim: the reference's command-line contract
// (d2q9-bgk.c:153-440) in front of liblbm_d2q9.so.
//
//   d2q9-bgk <paramfile> <obstaclefile>
//
// writes final_state.dat and av_vels.dat into the cwd (d2q9-bgk.c:63-64) and prints the five
// stdout lines of d2q9-bgk.c:411-415 byte-compatibly, followed by extra lines (MLUPS, roofline).
// Everything between tic and toc is the device path; this file only parses, times and writes.
// Environment (optional): LBM_DEVICE=<hip ordinal>, LBM_NO_OUTPUT=1 (like the reference's
// -DPROFILE build, :419-421), LBM_FLAGS=<lbm_create flags>.

#include <sys/resource.h>
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lbm_d2q9.h"

namespace {

[[noreturn]] void die(const char* message, int line, const char* file)   // d2q9-bgk.c:1145-1151
{
  std::fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  std::fprintf(stderr, "%s\n", message);
  std::fflush(stderr);
  std::exit(EXIT_FAILURE);
}

[[noreturn]] void usage(const char* exe)                                  // d2q9-bgk.c:1153-1157
{
  std::fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", exe);
  std::exit(EXIT_FAILURE);
}

double wall_seconds()
{
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + t.tv_usec / 1000000.0;
}

int env_int(const char* name, int dflt)
{
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

}  // namespace

int main(int argc, char* argv[])
{
  if (argc != 3) usage(argv[0]);                                           // :197-200

  lbm_params params;
  if (lbm_read_params(argv[1], &params)) die(lbm_last_error(), __LINE__, __FILE__);
  if (params.nx <= 0 || params.ny <= 0 || params.max_iters < 0) die("could not read param file: nx", __LINE__, __FILE__);
  std::vector<int> obstacles(static_cast<size_t>(params.nx) * params.ny);
  int free_cells = 0;
  if (lbm_read_obstacles(argv[2], params.nx, params.ny, obstacles.data(), &free_cells)) die(lbm_last_error(), __LINE__, __FILE__);

  lbm_ctx* ctx = nullptr;
  if (lbm_create(&ctx, &params, free_cells, obstacles.data(), 0, params.ny, env_int("LBM_DEVICE", 0),
                 static_cast<unsigned>(env_int("LBM_FLAGS", 0))))
    die(lbm_last_error(), __LINE__, __FILE__);
  std::vector<float> av_vels(static_cast<size_t>(params.max_iters) + 1);

  const double tic = wall_seconds();                                       // :278-279
  if (lbm_run(ctx, params.max_iters, av_vels.data())) die(lbm_last_error(), __LINE__, __FILE__);
  const double toc = wall_seconds();                                       // :397-398
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);                                             // :399-403
  const double usrtim = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1000000.0;
  const double systim = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1000000.0;

  std::vector<float> cells(static_cast<size_t>(params.nx) * params.ny * LBM_NSPEEDS);
  if (lbm_get_cells(ctx, cells.data())) die(lbm_last_error(), __LINE__, __FILE__);
  const float free_cells_inv = 1.0f / free_cells;                          // :950
  const float av = lbm_av_velocity_host(&params, cells.data(), obstacles.data(), params.ny) * free_cells_inv;   // :753
  std::printf("==done==\n");                                               // :411-415
  std::printf("Reynolds number:\t\t%.12E\n", lbm_reynolds(&params, av));
  std::printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  std::printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usrtim);
  std::printf("Elapsed system CPU time:\t%.6lf (s)\n", systim);
  const double mlups = static_cast<double>(params.nx) * params.ny * params.max_iters / (toc - tic) / 1e6;
  std::printf("MLUPS:\t\t\t\t%.1f\n", mlups);
  std::printf("HBM roofline (108 B/cell-step @ 8.0 TB/s = 74074 MLUPS):\t%.1f %%\n", 100.0 * mlups / (8.0e12 / 108.0 / 1e6));

  if (!env_int("LBM_NO_OUTPUT", 0)) {                                      // :419-421
    if (lbm_write_final_state("final_state.dat", &params, cells.data(), obstacles.data(), params.ny, 0, 0))
      die(lbm_last_error(), __LINE__, __FILE__);
    if (lbm_write_av_vels("av_vels.dat", av_vels.data(), params.max_iters)) die(lbm_last_error(), __LINE__, __FILE__);
  }
  lbm_destroy(ctx);
  return EXIT_SUCCESS;                                                     // :439
}
