/* datconv -- convert a data or code file between the text format of SOM_PAK / LVQ_PAK (datafile.c:396-447,
 * 552-748) and the raw fp32 side format every tool of this package also reads ("#!somf32", paklib.c), or
 * materialise the seeded generator stream (-din gen:k=..,dim=..,n=..,seed=..) as a file.  No GPU involved.
 * at 40 MB/s the text parser, not the engine, is the wall for anything beyond C3. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "datconv - text <-> raw fp32 conversion of .dat/.cod files\n"
    "Required:  -din file|gen:spec   -dout file\n"
    "Optional:  -text   write the text format (default: raw fp32)\n"
    "           -noskip keep rows whose components are all masked\n";

int main(int argc, char **argv)
{
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); return 0; }
  char *din = extract_parameter(argc, argv, "-din", ALWAYS), *dout = extract_parameter(argc, argv, "-dout", ALWAYS);
  int text = extract_parameter(argc, argv, "-text", OPTION2) != NULL;
  int noskip = extract_parameter(argc, argv, "-noskip", OPTION2) != NULL;
  struct entries *e = open_entries(din, 0, !noskip);
  if (!e) { fprintf(stderr, "Can't open data file '%s'\n", din); return 1; }
  ifverbose(1) fprintf(stderr, "%ld rows of %d components -> %s (%s)\n", e->num_entries, e->dimension, dout, text ? "text" : "raw fp32");
  int rc = text ? save_entries(e, dout) : save_entries_f32(e, dout);
  close_entries(e);
  return rc ? 1 : 0;
}
