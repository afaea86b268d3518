/* eveninit / propinit -- initial LVQ codebook picked from the data (LVQ_PAK eveninit.c:46-234,
 * pick_inside_codes lvq_rout.c:137-195): the same number of codes per class (eveninit) or a
 * number proportional to the class size (propinit), taking in file order only entries that a
 * k-NN vote over the whole data set classifies correctly.  The O(n^2 d) part -- the k nearest
 * neighbours of every entry -- runs on the MI355X engine in one pass; the picking itself is the
 * reference's sequential logic on the host. */
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "pak.h"

static const char *usage =
    "eveninit/propinit - initial codebook for LVQ (MI355X engine)\n"
    "Required:  -din file  -cout file  -noc N\n"
    "Optional:  -type eveninit|propinit  -knn N (default 5, at most 8)  -rand seed  -v level\n";

/* pick_inside_codes: at most need[c] entries of each class, from the beginning of the data,
 * that are correctly classified by k-NN; appends row indices to out */
static long pick_inside(struct entries *data, const unsigned char *ok, struct hitlist *classes, long *out)
{
  long total = 0, n = 0;
  for (long c = 0; c < classes->entries; c++) total += classes->freq[c];
  for (long r = 0; total && r < data->num_entries; r++) {
    int lab = get_entry_label(&data->rows[r]);
    long c;
    for (c = 0; c < classes->entries; c++) if (classes->label[c] == lab) break;
    if (c < classes->entries && classes->freq[c] > 0 && ok[r]) {
      total--;
      out[n++] = r;
      classes->freq[c]--;
    }
  }
  return n;
}

int main(int argc, char **argv)
{
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  const char *progname = pak_progname(argv[0]);
  int prop = -1;
  if (strcasecmp(progname, "propinit") == 0) prop = 1;
  else if (strcasecmp(progname, "eveninit") == 0) prop = 0;
  char *pname = extract_parameter(argc, argv, "-type", prop < 0 ? ALWAYS : OPTION);
  if (pname) {
    if (strcasecmp(pname, "propinit") == 0) prop = 1;
    else if (strcasecmp(pname, "eveninit") == 0) prop = 0;
  }
  if (prop < 0) { fprintf(stderr, "unknown init type\n"); exit(1); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  long number_of_codes = (int)oatoi(extract_parameter(argc, argv, "-noc", ALWAYS), 1);
  int knn = (int)oatoi(extract_parameter(argc, argv, "-knn", OPTION), 5);
  int randomize = (int)oatoi(extract_parameter(argc, argv, "-rand", OPTION), 0);

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  init_random(randomize);

  /* init_codes, eveninit.c:46-158 */
  struct hitlist *classes = new_hitlist();
  for (long r = 0; r < data->num_entries; r++) add_hit(classes, get_entry_label(&data->rows[r]));
  long nol = classes->entries, tot = data->num_entries;
  if (nol > number_of_codes) fprintf(stderr, "There are more different classes than requested codes");
  long nic = nol ? number_of_codes / nol : 0;
  ifverbose(2) fprintf(stderr, "The codebook vectors for each class are picked\n");
  long want = 0;
  for (long c = 0; c < nol; c++) {
    if (prop) {
      classes->freq[c] = (long)((float)classes->freq[c] * (float)number_of_codes / tot);
      if (classes->freq[c] < 1) classes->freq[c] = 1;
    } else {
      classes->freq[c] = nic;
    }
    want += classes->freq[c];
  }
  unsigned char *ok = knn_correct_all(data, knn);
  if (!ok) { fprintf(stderr, "Failed to initialize codes\n"); exit(1); }
  long cap = want + number_of_codes + 16;
  long *picked = malloc(sizeof(long) * cap);
  long nom = pick_inside(data, ok, classes, picked);
  long emp = 0;
  for (long c = 0; c < nol; c++) if (classes->freq[c] == 0) emp++;
  ifverbose(2) fprintf(stderr, "For %ld classes all found\n", emp);
  ifverbose(2) fprintf(stderr, "Found %ld vectors in first pass\n", nom);
  if (nom < number_of_codes) {
    float frac = 0.0, err = 0.0;
    if (emp != 0) frac = (number_of_codes - nom) / (float)emp;
    for (long c = 0; c < nol; c++) {
      if (classes->freq[c] == 0) {
        classes->freq[c] = (int)(frac + err);
        err = frac + err - classes->freq[c];
      } else {
        classes->freq[c] = 0;
      }
    }
    /* pick more codes from those classes where you got all (from the beginning again) */
    nom += pick_inside(data, ok, classes, picked + nom);
  }
  struct entries *codes = pick_rows(data, picked, nom);
  codes->topol = TOPOL_LVQ;
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
  invalidate_alphafile(out_code_file);
  free(ok); free(picked); free_hitlist(classes);
  close_entries(codes); close_entries(data);
  pak_shutdown();
  return 0;
}
