/* balance -- even out the codebook by the medians of the shortest within-class distances
 * (LVQ_PAK balance.c:44-283): classes whose codes sit close together lose one, classes whose
 * codes are far apart gain one (picked from the data like eveninit does), then one OLVQ1 pass
 * over the data redistributes the codes.  The two heavy parts -- the k-NN vote behind the picking
 * and olvq1_training -- run on the MI355X engine; the medians are O(sum n_c^2 d) over the
 * codebook and stay on the host. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pak.h"

#define BAL 1.3

static const char *usage =
    "balance - balances the number of entries in codebook by shortest distances (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file\n"
    "Optional:  -knn N (default 5, at most 8)  -rand seed  -selfuncs hip  -v level\n";

/* vector_dist_euc, lvq_pak.c:291-316 */
static float dist_euc(const struct data_entry *a, const struct data_entry *b, int dim)
{
  float diff, difference = 0.0;
  int masked = 0;
  for (int i = 0; i < dim; i++) {
    if ((a->mask && a->mask[i]) || (b->mask && b->mask[i])) masked++;
    else { diff = a->points[i] - b->points[i]; difference += diff * diff; }
  }
  if (masked == dim) return -1;
  return sqrt(difference);
}

static int cmp_float(const void *a, const void *b)
{
  float x = *(const float *)a, y = *(const float *)b;
  return x < y ? -1 : x > y ? 1 : 0;
}

struct mindists { long num_classes; long *cls; long *noe; float *dists; };

/* med_distances, lvq_rout.c:373-491: per class (most frequent first), the median over its entries
 * of the distance to the nearest LATER entry of the same class */
static struct mindists *med_distances(struct entries *codes)
{
  struct mindists *md = calloc(1, sizeof *md);
  struct hitlist *classes = new_hitlist();
  int dim = codes->dimension;
  for (long r = 0; r < codes->num_entries; r++) add_hit(classes, get_entry_label(&codes->rows[r]));
  long nol = classes->entries;
  md->num_classes = nol;
  md->cls = calloc(nol + 1, sizeof(long)); md->noe = calloc(nol + 1, sizeof(long)); md->dists = calloc(nol + 1, sizeof(float));
  long mnoe = nol ? classes->freq[0] : 0;
  float *meds = malloc(sizeof(float) * (mnoe + 1));
  for (long i = 0; i < nol; i++) {
    md->cls[i] = classes->label[i];
    md->noe[i] = classes->freq[i];
    long not = 0;
    for (long r = 0; r < codes->num_entries; r++) {
      if (get_entry_label(&codes->rows[r]) != md->cls[i]) continue;
      float dissf = FLT_MAX;
      int fou = 0;
      for (long s = r + 1; s < codes->num_entries; s++)
        if (get_entry_label(&codes->rows[s]) == md->cls[i]) {
          fou = 1;
          float dist = dist_euc(&codes->rows[s], &codes->rows[r], dim);
          if (dist < dissf) dissf = dist;
        }
      if (fou) meds[not++] = dissf;
    }
    if (not > 0) { qsort(meds, not, sizeof(float), cmp_float); md->dists[i] = meds[not / 2]; }
  }
  free(meds); free_hitlist(classes);
  return md;
}
static void free_mindists(struct mindists *md) { if (md) { free(md->cls); free(md->noe); free(md->dists); free(md); } }

/* a + b as one new block (copies) */
static struct entries *join_entries(struct entries *a, const long *arows, long na, struct entries *b,
                                    const long *brows, long nb)
{
  struct entries *pa = pick_rows(a, arows, na), *pb = pick_rows(b, brows, nb);
  long n = na + nb;
  long *all = malloc(sizeof(long) * (n + 1));
  for (long k = 0; k < n; k++) all[k] = k;
  /* gather through a temporary that views both */
  struct entries tmp = *pa;
  tmp.num_entries = n;
  tmp.rows = malloc(sizeof(struct data_entry) * (n + 1));
  memcpy(tmp.rows, pa->rows, sizeof(struct data_entry) * na);
  memcpy(tmp.rows + na, pb->rows, sizeof(struct data_entry) * nb);
  tmp.masks = (pa->masks || pb->masks) ? (char *)1 : NULL;       /* only tested for non-NULL by pick_rows */
  struct entries *out = pick_rows(&tmp, all, n);
  free(tmp.rows); free(all);
  close_entries(pa); close_entries(pb);
  return out;
}

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  int knn = (int)oatoi(extract_parameter(argc, argv, "-knn", OPTION), 5);
  int randomize = (int)oatoi(extract_parameter(argc, argv, "-rand", OPTION), 0);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 1, 1);
  if (!codes) { fprintf(stderr, "Can't open code file '%s'\n", in_code_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codes have different dimensions\n");
    close_entries(codes); close_entries(data); exit(1);
  }
  init_random(randomize);

  /* balance_codes, balance.c:44-226 */
  ifverbose(2) fprintf(stderr, "Medians of the shortest distances are computed\n");
  struct mindists *md = med_distances(codes);
  long nol = md->num_classes;
  long *noe = md->noe, *cls = md->cls;
  float *dists = md->dists;
  int *diff = calloc(nol + 1, sizeof(int));
  float aver = 0.0;
  int note = 0;
  for (long i = 0; i < nol; i++) if (noe[i] > 1) { aver += dists[i]; note++; }
  aver /= note;
  note = 0;
  ifverbose(2) fprintf(stderr, "Medians of different classes are compared\n");
  for (long i = 0; i < nol; i++) {
    if ((aver > BAL * dists[i]) && (noe[i] > 1)) { diff[i]--; note++; }
    if (BAL * aver < dists[i]) { diff[i]++; note--; }
  }
  /* (classes come from the codebook itself, so none has noe == 0: the reference's force-pick
   * branch, balance.c:107-119, cannot trigger) */
  for (long i = 0; i < nol; i++) {
    if ((aver > BAL * dists[i]) && ((noe[i] + diff[i]) > 1)) { if (note < 0) { diff[i]--; note++; } }
    if (BAL * aver < dists[i]) { if (note > 0) { diff[i]++; note--; } }
  }
  ifverbose(1) fprintf(stderr, "Some codebook vectors are removed\n");
  long *keep = malloc(sizeof(long) * (codes->num_entries + 1)), nkeep = 0;
  for (long r = 0; r < codes->num_entries; r++) {
    int label = get_entry_label(&codes->rows[r]);
    long i;
    for (i = 0; i < nol; i++) if (cls[i] == label) break;
    if (diff[i] < 0) diff[i]++;
    else keep[nkeep++] = r;
  }
  ifverbose(1) fprintf(stderr, "Some new codebook vectors are picked\n");
  struct hitlist *more = new_hitlist();
  long total = 0;
  for (long i = 0; i < nol; i++)
    while (diff[i] > 0) { add_hit(more, cls[i]); diff[i]--; total++; }
  long *picked = malloc(sizeof(long) * (total + 1)), npicked = 0;
  if (total > 0) {                                    /* pick_inside_codes, lvq_rout.c:137-195 */
    unsigned char *ok = knn_correct_all(data, knn);
    if (!ok) exit(1);
    long left = total;
    for (long r = 0; left && r < data->num_entries; r++) {
      int lab = get_entry_label(&data->rows[r]);
      long c;
      for (c = 0; c < more->entries; c++) if (more->label[c] == lab) break;
      if (c < more->entries && more->freq[c] > 0 && ok[r]) { left--; picked[npicked++] = r; more->freq[c]--; }
    }
    free(ok);
  }
  free_hitlist(more);
  struct entries *newcodes = join_entries(codes, keep, nkeep, data, picked, npicked);
  newcodes->topol = codes->topol; newcodes->neigh = codes->neigh; newcodes->xdim = codes->xdim; newcodes->ydim = codes->ydim;
  const long nkeep_total = nkeep, npicked_total = npicked;
  free(keep); free(picked); free(diff);
  close_entries(codes);
  codes = newcodes;

  ifverbose(1) fprintf(stderr, "Codebook vectors are redistributed\n");
  set_teach_params(&teach, codes, data, funcname);
  teach.knn = knn;
  teach.length = data->num_entries;
  teach.alpha = 0.3;
  struct entries *red = olvq1_training(&teach, NULL, out_code_file);
  if (!red) exit(1);
  /* The reference never counts the entries it has just appended ("laske montako uutta",
   * balance.c:188): olvq1_training sees num_entries = kept codes only, so its learning-rate file
   * holds one line per KEPT code, and the rates of the appended ones are read and written past the
   * end of its array (heap overflow: the reference's result then depends on the allocator, or it
   * aborts).  Here every code starts at the 0.3 the reference intends (balance.c:203); the file
   * keeps the reference's length.  Inputs where the appended codes only win their own sample --
   * the cases that are stable in the reference -- give identical bytes (tests/test_cli_tools.py). */
  if (npicked_total > 0) {
    float *ta = malloc(sizeof(float) * (nkeep_total + 1));
    if (alpha_read(ta, nkeep_total, out_code_file)) alpha_write(ta, nkeep_total, out_code_file);
    free(ta);
  }

  ifverbose(2) fprintf(stderr, "Medians of the shortest distances are computed\n");
  free_mindists(md);
  md = med_distances(red);
  verbose_level = 1;                                   /* `if (verbose(1) > 0)` at balance.c:214 sets the level */
  for (long i = 0; i < md->num_classes; i++)
    fprintf(stdout, "In class %9s %3d units, min dist.: %.3f\n", find_conv_to_lab((int)md->cls[i]), (int)md->noe[i],
            md->dists[i]);
  free_mindists(md);
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(red, out_code_file);
  close_entries(red); close_entries(data);
  pak_shutdown();
  return 0;
}
