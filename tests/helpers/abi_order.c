/* abi_order.c -- a plain C host that tears the handles down in the "wrong" order: engine first, then its
 * codebook and data set.  include/somhip.h promises that nothing aborts and that any order of the destroy calls
 * is fine; calls on an orphaned handle must fail with a message.  Exit status 0 = all of that held. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "somhip.h"

#define EXPECT(cond, what) do { if (!(cond)) { fprintf(stderr, "FAILED: %s (%s)\n", what, somhip_last_error()); return 1; } } while (0)

int main(void)
{
  enum { N = 130, D = 9, M = 70 };
  static float rows[N * D], data[M * D], back[N * D], diff[M];
  static int32_t idx[M];
  for (int i = 0; i < N * D; i++) rows[i] = (float)((i * 37) % 101) / 7.0f;
  for (int i = 0; i < M * D; i++) data[i] = (float)((i * 53) % 89) / 5.0f;
  for (int round = 0; round < 2; round++) {
    somhip_engine *e = NULL;
    somhip_codebook *cb = NULL;
    somhip_dataset *ds = NULL;
    EXPECT(somhip_engine_create(0, &e) == 0, "engine_create");
    EXPECT(somhip_codebook_create(e, rows, NULL, N, D, SOMHIP_TOPOL_LVQ, 0, 0, 0, 0, N, &cb) == 0, "codebook_create");
    EXPECT(somhip_dataset_create(e, data, M, D, NULL, NULL, NULL, NULL, &ds) == 0, "dataset_create");
    EXPECT(somhip_find_winners(cb, ds, 0, M, 1, SOMHIP_TIE_FIRST, idx, diff, NULL) == 0, "find_winners");
    if (round == 0) {                      /* engine first */
      somhip_engine_destroy(e);
      EXPECT(somhip_find_winners(cb, ds, 0, M, 1, SOMHIP_TIE_FIRST, idx, diff, NULL) != 0, "find_winners on orphans must fail");
      EXPECT(strstr(somhip_last_error(), "destroyed") != NULL, "message names the destroyed engine");
      EXPECT(somhip_codebook_download(cb, back) != 0, "download from an orphan must fail");
      somhip_codebook_destroy(cb);
      somhip_dataset_destroy(ds);
    } else {                               /* children first (the documented order) */
      EXPECT(somhip_codebook_download(cb, back) == 0 && memcmp(back, rows, sizeof rows) == 0, "download");
      somhip_dataset_destroy(ds);
      somhip_codebook_destroy(cb);
      somhip_engine_destroy(e);
    }
  }
  /* bad arguments come back as errors, not as crashes */
  somhip_engine *e = NULL;
  somhip_codebook *cb = NULL;
  EXPECT(somhip_engine_create(0, &e) == 0, "engine_create");
  EXPECT(somhip_codebook_create(e, rows, NULL, 0, D, SOMHIP_TOPOL_LVQ, 0, 0, 0, 0, 0, &cb) != 0, "empty codebook refused");
  EXPECT(somhip_codebook_create(e, rows, NULL, N, D, SOMHIP_TOPOL_HEXA, 1, 7, 7, 0, N, &cb) != 0, "map of the wrong size refused");
  somhip_engine_destroy(e);
  somhip_engine_destroy(NULL);
  somhip_codebook_destroy(NULL);
  somhip_dataset_destroy(NULL);
  printf("abi_order ok\n");
  return 0;
}
