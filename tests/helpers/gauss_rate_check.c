/* gauss_rate_check.c -- the fast form of the gaussian rate (som_lvq_pak_amd/csrc/kernels/gauss_rate.hpp, compiled for
 * the host) against the chain as the reference writes it (som_rout.c:539-542, hexa_dist :438-451) with glibc's sqrt
 * and exp: every argument the fast form decides must give the same float; prints the counts.
 *   gauss_rate_check <cases> <seed>      exit 1 on a mismatch */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gauss_rate.hpp"

static uint64_t s;
static uint32_t rnd(void) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }

int main(int argc, char **argv) {
  long cases = argc > 1 ? atol(argv[1]) : 1000000, decided = 0, zero = 0, bad = 0, badq = 0;
  s = argc > 2 ? (uint64_t)atol(argv[2]) : 1;
  for (long i = 0; i < cases; i++) {
    float diff = (float)(rnd() % 700) * 0.5f, dy = (float)(rnd() % 400), lat, radius;
    uint32_t kind = rnd() % 8;
    lat = diff * diff; lat += 0.75f * dy * dy;
    if (kind == 0) { uint32_t b = 0x3f800000u + rnd() % 0x04000000u; memcpy(&radius, &b, 4); }      /* [1, 256): any mantissa */
    else if (kind == 1) radius = 1.0f + (float)(rnd() % 1000) * 0.001f;
    else if (kind == 2) { uint32_t b = rnd(); memcpy(&radius, &b, 4); radius = fabsf(radius); if (!(radius == radius)) radius = 0.0f; }  /* anything, 0 and inf included */
    else radius = 1.0f + 127.0f * (float)(rnd() % 10000001) / 1e7f;
    if (kind == 3) lat = (float)(rnd() % 64) * 0.25f;                                                  /* near the winner */
    /* the reference's chain */
    float dd = (float)sqrt((double)lat);
    double yref = (double)(-dd * dd) / (2.0 * radius * radius);
    float want = (float)exp(yref);
    double den, rcp;
    float got;
    gauss_rate_den(radius, &den, &rcp);
    if (rcp == rcp) {                                   /* the quotient on its own */
      double n = (double)(-dd * dd), q = n * rcp, r = fma(-q, den, n), y = fma(r, rcp, q);
      if (memcmp(&y, &yref, 8) != 0 && !(y == 0.0 && yref == 0.0)) { if (badq++ < 5) fprintf(stderr, "quotient: lat %a radius %a: %a vs %a\n", lat, radius, y, yref); }
    }
    if (gauss_rate_fast(lat, den, rcp, &got)) {
      decided++;
      if (got == 0.0f) zero++;
      if (memcmp(&got, &want, 4) != 0) { if (bad++ < 5) fprintf(stderr, "rate: lat %a radius %a: %a vs %a\n", lat, radius, got, want); }
    }
  }
  printf("gauss_rate_check: %ld cases, %ld decided by the fast form (%ld of them zero), %ld left to the library chain, %ld wrong floats, %ld wrong quotients\n",
         cases, decided, zero, cases - decided, bad, badq);
  return bad || badq ? 1 : 0;
}
