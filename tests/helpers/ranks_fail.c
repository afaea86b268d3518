/* ranks_fail.c -- pak_run_ranks (paklib.c) when a rank dies: rank 1 fails at once (mode "exit": returns 1; "signal":
 * raises SIGSEGV) while the other ranks block for ever -- rank 0 in a read from a rank that is still alive (the socket
 * of a dead peer would return EOF; a rank waiting inside a collective of RCCL sees nothing of the kind), the rest in
 * pause().  The call must come back with 1 within seconds and leave no child behind; mode "ok": all ranks return 0.
 * CPU only: no rank touches a GPU.  Prints "returned R after S s; children left N". */
#define _GNU_SOURCE
#include <errno.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "pak.h"
static const char *mode;
static int rank_main(int rank, int world, int *fds, void *arg)
{
  (void)arg;
  if (!strcmp(mode, "ok")) {               /* every rank tells rank 0 its number */
    if (rank == 0) { int sum = 0; for (int r = 1; r < world; r++) { int v; if (pak_sock_read(fds[r - 1], &v, sizeof v)) return 1; sum += v; } return sum != world * (world - 1) / 2; }
    return pak_sock_write(fds[0], &rank, sizeof rank);
  }
  if (rank == 1) { if (!strcmp(mode, "signal")) raise(SIGSEGV); return 1; }
  if (rank == 0) { char c; pak_sock_read(fds[world - 2], &c, 1); return 0; }   /* from the last rank, which only pauses */
  for (;;) pause();
}
int main(int argc, char **argv)
{
  mode = argc > 1 ? argv[1] : "exit";
  const int world = argc > 2 ? atoi(argv[2]) : 3;
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  const int rc = pak_run_ranks(world, rank_main, NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  int left = 0;
  while (waitpid(-1, NULL, WNOHANG) >= 0) left++;      /* ECHILD at once when every rank was reaped */
  printf("returned %d after %.1f s; children left %d\n", rc, (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec), left);
  return 0;
}
