/* pak_parse_float against sscanf("%f") on generated tokens: prints the number of mismatches. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pak.h"
static unsigned long long rs = 88172645463325252ULL;
static unsigned long long rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
int main(int argc, char **argv)
{
  long n = argc > 1 ? atol(argv[1]) : 2000000, bad = 0, fast = 0;
  char tok[96];
  const char *special[] = {"0", "-0", "+0.0", ".5", "5.", "-.25e1", "1e", "1e+", "abc", "12abc", "0x10", "inf", "nan", "1e-45",
                           "3.4028235e38", "3.5e38", "1e39", "1.17549435e-38", "1e-40", "16777217", "16777216.5",
                           "0.1", "123456789012345678901234567890", "1.00000000000000000000001", "9007199254740993",
                           "1.0000000596046448", "1.00000005960464477539", "1.000000059604644775390625", "8.5", "33554434.0000001"};
  for (unsigned k = 0; k < sizeof special / sizeof *special; k++) {
    float a = -777.f, b = -777.f;
    int ra = pak_parse_float(special[k], &a), rb = sscanf(special[k], "%f", &b) > 0;
    if (ra != rb || (ra && memcmp(&a, &b, 4))) { bad++; printf("special '%s': %d %a vs %d %a\n", special[k], ra, a, rb, b); }
  }
  for (long i = 0; i < n; i++) {
    int kind = (int)(rnd() % 6);
    if (kind == 0) snprintf(tok, sizeof tok, "%g", (double)(float)((double)(long long)(rnd() % 2000001 - 1000000) / 1000.0));
    else if (kind == 1) { float f; unsigned u = (unsigned)rnd(); memcpy(&f, &u, 4); if (f != f || f - f != 0) f = 1.5f; snprintf(tok, sizeof tok, "%.9g", f); }
    else if (kind == 2) snprintf(tok, sizeof tok, "%llu.%llu", rnd() % 100000, rnd() % 1000000000);
    else if (kind == 3) snprintf(tok, sizeof tok, "%s%llue%d", rnd() & 1 ? "-" : "", rnd() % 1000000000000ULL, (int)(rnd() % 60) - 30);
    else if (kind == 4) { /* straddle a float tie */
      float f; unsigned u = 0x3f800000u + (unsigned)(rnd() % 8000000); memcpy(&f, &u, 4);
      double mid = (double)f + 0.5 * ((double)__builtin_nextafterf(f, 2.0f * f) - (double)f);
      snprintf(tok, sizeof tok, "%.*g", 17 + (int)(rnd() % 3), mid + ((int)(rnd() % 3) - 1) * 1e-16);
    } else snprintf(tok, sizeof tok, "%.6f", (double)(long long)(rnd() % 20000001 - 10000000) / 997.0);
    float a = 0, b = 0;
    int ra = pak_parse_float(tok, &a), rb = sscanf(tok, "%f", &b) > 0;
    if (ra != rb || memcmp(&a, &b, 4)) { if (bad < 10) printf("'%s': %a vs %a\n", tok, a, b); bad++; }
  }
  (void)fast;
  printf("mismatches %ld\n", bad);
  return bad != 0;
}
