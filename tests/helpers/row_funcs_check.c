/* row_funcs_check.c -- the per-sample dist / vector_adapt of the "hip" registry row (paklib.c) against the oracle's
 * restatement of vector_dist_euc / adapt_vector (lvq_pak.c:291-316, 339-351) on generated rows with masks: bits.
 * CPU only (set_teach_params does not touch the GPU).  Prints "mismatches N". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pak.h"
#include "som_lvq_oracle.h"
static unsigned long long rs = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static float frnd(void) { return (float)((double)(long long)(rnd() % 2000001 - 1000000) / 3171.0); }
int main(void)
{
  struct entries codes; memset(&codes, 0, sizeof codes);
  struct teach_params tp; memset(&tp, 0, sizeof tp);
  codes.topol = TOPOL_LVQ;
  set_teach_params(&tp, &codes, NULL, "hip");
  if (!tp.dist || !tp.vector_adapt || !tp.winner) { printf("registry row incomplete\n"); return 2; }
  long bad = 0;
  for (int it = 0; it < 20000; it++) {
    int d = 1 + (int)(rnd() % 40);
    float a[40], b[40], c[40];
    char ma[40], mb[40];
    for (int i = 0; i < d; i++) { a[i] = frnd(); b[i] = frnd(); ma[i] = rnd() % 5 == 0; mb[i] = rnd() % 7 == 0; }
    int kind = it % 4;                       /* no masks, one side, both, everything masked */
    if (kind == 3) for (int i = 0; i < d; i++) ma[i] = 1;
    struct data_entry ea = { a, NULL, 0, 0, kind >= 1 ? ma : NULL, NULL }, eb = { b, NULL, 0, 0, kind >= 2 ? mb : NULL, NULL };
    float got = tp.dist(&ea, &eb, d);
    float want = orc_vector_dist_euc(a, (unsigned char *)ea.mask, b, (unsigned char *)eb.mask, d);
    if (memcmp(&got, &want, 4)) { if (bad < 5) printf("dist %a vs %a (d %d kind %d)\n", got, want, d, kind); bad++; }
    float alpha = (float)((double)(long long)(rnd() % 2001 - 1000) / 997.0);      /* negative: LVQ push-away */
    memcpy(c, a, sizeof a);
    tp.vector_adapt(&ea, &eb, d, alpha);     /* code = a (its own mask is ignored), sample = b with b's mask */
    orc_adapt_vector(c, b, (unsigned char *)eb.mask, d, alpha);
    if (memcmp(a, c, sizeof(float) * d)) { if (bad < 5) printf("adapt differs (d %d kind %d)\n", d, kind); bad++; }
  }
  printf("mismatches %ld\n", bad);
  return bad != 0;
}
