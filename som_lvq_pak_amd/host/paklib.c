/* paklib.c -- host library of the tools: arguments, labels, hit lists, .dat/.cod text files,
 * the -rand shuffle, the -selfuncs registry and the epoch-level functions that hand the
 * hot path to libsomhip.so.  Written from scratch over dense storage; file formats, flag
 * names, messages and numerics follow SOM_PAK/LVQ_PAK 3.2 (citations: file:line in
 * hynde/som_lvq_pak). */
#define _GNU_SOURCE
#include "pak.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------ arguments */
int verbose_level = 1;
int use_fixed_level = 0, use_weights_level = 0;

/* lvq_pak.c:583-612: linear search, the value is the next argv; OPTION2 flags take none */
char *extract_parameter(int argc, char **argv, const char *param, int when)
{
  int i = 0;
  while (i < argc && strcmp(param, argv[i]) != 0) i++;
  if (i <= argc - 1 && when == OPTION2) return "";
  if (i < argc - 1) return argv[i + 1];
  if (when == ALWAYS) {
    fprintf(stderr, "Can't find asked option %s\n", param);
    exit(-1);
  }
  return NULL;
}
long oatoi(const char *s, long def) { return s ? atol(s) : def; }
float oatof(const char *s, float def) { return s ? (float)atof(s) : def; }

static const char *masked_string = "x";          /* datafile.h:33, -mask_str / LVQSOM_MASK_STR */

int global_options(int argc, char **argv)        /* lvq_pak.c:618-660 */
{
  char *s = getenv("LVQSOM_MASK_STR");
  if (s) masked_string = s;
  s = extract_parameter(argc, argv, "-mask_str", OPTION);
  if (s) masked_string = s;
  if (extract_parameter(argc, argv, "-version", OPTION2))
    fprintf(stderr, "Version: som_lvq_pak_amd (MI355X engine, libsomhip %d), file formats of SOM/LVQ_PAK 3.2\n",
            somhip_version());
  verbose_level = (int)oatoi(extract_parameter(argc, argv, "-v", OPTION), 1);
  return 0;
}

const char *pak_progname(const char *argv0)      /* fileio.c:433-465: basename of argv[0] */
{
  const char *p = strrchr(argv0, '/');
  return p ? p + 1 : argv0;
}

/* ------------------------------------------------------------------ labels */
static char **label_names = NULL;                /* [0] unused: 0 = LABEL_EMPTY */
static int label_count = 0, label_cap = 0;

int find_conv_to_ind(const char *str)
{
  for (int i = 1; i <= label_count; i++)
    if (strcmp(label_names[i], str) == 0) return i;
  if (label_count + 2 > label_cap) {
    label_cap = label_cap ? 2 * label_cap : 64;
    label_names = realloc(label_names, sizeof(char *) * label_cap);
  }
  label_names[++label_count] = strdup(str);
  return label_count;
}
const char *find_conv_to_lab(int ind) { return (ind >= 1 && ind <= label_count) ? label_names[ind] : NULL; }
int number_of_labels(void) { return label_count; }

struct hitlist *new_hitlist(void) { return calloc(1, sizeof(struct hitlist)); }
void free_hitlist(struct hitlist *h) { if (h) { free(h->label); free(h->freq); free(h); } }
long add_hit(struct hitlist *h, long label)
{
  long i;
  for (i = 0; i < h->entries; i++) if (h->label[i] == label) break;
  if (i == h->entries) {
    if (h->entries == h->cap) {
      h->cap = h->cap ? 2 * h->cap : 8;
      h->label = realloc(h->label, sizeof(long) * h->cap);
      h->freq = realloc(h->freq, sizeof(long) * h->cap);
    }
    h->label[i] = label; h->freq[i] = 1; h->entries++;
    return 1;
  }
  long f = ++h->freq[i];
  while (i > 0 && h->freq[i - 1] < f) {          /* strictly smaller: ties keep their order */
    long tl = h->label[i - 1], tf = h->freq[i - 1];
    h->label[i - 1] = h->label[i]; h->freq[i - 1] = h->freq[i];
    h->label[i] = tl; h->freq[i] = tf;
    i--;
  }
  return f;
}
long hitlist_label_freq(struct hitlist *h, long label)
{
  for (long i = 0; i < h->entries; i++) if (h->label[i] == label) return h->freq[i];
  return 0;
}

/* ------------------------------------------------------------------ files */
static const char *topol_names[] = {NULL, "data", "lvq", "hexa", "rect"};
static const char *neigh_names[] = {NULL, "bubble", "gaussian"};

static int id_of(const char **names, int n, const char *s)
{
  if (s) for (int i = 1; i < n; i++) if (strcasecmp(names[i], s) == 0) return i;
  return 0;
}

static FILE *open_text(const char *name, const char *mode, int *is_pipe)
{
  size_t len = strlen(name);
  *is_pipe = 0;
  if (strcmp(name, "-") == 0) return mode[0] == 'r' ? stdin : stdout;
  int gz = (len > 3 && strcmp(name + len - 3, ".gz") == 0) ||
           (len > 2 && (strcmp(name + len - 2, ".z") == 0 || strcmp(name + len - 2, ".Z") == 0));
  if (gz) {                                      /* fileio.c:57-200: compressed files through gzip */
    char cmd[4096];
    snprintf(cmd, sizeof cmd, mode[0] == 'r' ? "gzip -d -c %s" : "gzip -9 -c >%s", name);
    *is_pipe = 1;
    return popen(cmd, mode[0] == 'r' ? "r" : "w");
  }
  return fopen(name, mode);
}
static void close_text(FILE *fp, int is_pipe)
{
  if (fp == stdin || fp == stdout) return;
  if (is_pipe) pclose(fp); else fclose(fp);
}

void clear_entry_labels(struct entries *e, long r)
{
  free(e->rows[r].labels);
  e->rows[r].labels = NULL;
  e->rows[r].num_labs = 0;
}
void add_entry_label(struct entries *e, long r, int label)
{
  struct data_entry *d = &e->rows[r];
  d->labels = realloc(d->labels, sizeof(int) * (d->num_labs + 1));
  d->labels[d->num_labs++] = label;
}

/* One vector component, with the value sscanf("%f") gives (the reference's load_entry,
 * datafile.c:627, 664): correctly rounded to float.  Plain decimals -- at most 19 significant
 * digits, mantissa below 2^53, |power of ten| <= 22 -- are formed with ONE double operation
 * (m * 10^e or m / 10^e, both operands exact, so the double is the correctly rounded value) and
 * then narrowed; the only way the second rounding can go wrong is a double that sits exactly on a
 * float tie, and that case, like everything unusual (hex, inf/nan, long digit strings, trailing
 * characters, float under/overflow), goes to sscanf itself.  ~10x faster than sscanf per token. */
int pak_parse_float(const char *s, float *out)
{
  static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14,
                                 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *p = s;
  int neg = 0, nd = 0, any = 0, e10 = 0;
  unsigned long long m = 0;
  if (*p == '-') { neg = 1; p++; } else if (*p == '+') p++;
  for (; *p >= '0' && *p <= '9'; p++) {
    any = 1;
    if (nd == 0 && *p == '0') continue;
    if (nd >= 19) goto slow;
    m = m * 10 + (unsigned)(*p - '0'); nd++;
  }
  if (*p == '.') {
    p++;
    for (; *p >= '0' && *p <= '9'; p++) {
      any = 1;
      e10--;
      if (nd == 0 && *p == '0') continue;
      if (nd >= 19) goto slow;
      m = m * 10 + (unsigned)(*p - '0'); nd++;
    }
  }
  if (!any) goto slow;
  if (*p == 'e' || *p == 'E') {
    p++;
    int eneg = 0, ex = 0, ed = 0;
    if (*p == '-') { eneg = 1; p++; } else if (*p == '+') p++;
    for (; *p >= '0' && *p <= '9'; p++) { if (ex < 10000) ex = ex * 10 + (*p - '0'); ed++; }
    if (!ed) goto slow;
    e10 += eneg ? -ex : ex;
  }
  if (*p != '\0') goto slow;
  if (m == 0) { *out = neg ? -0.0f : 0.0f; return 1; }
  if (m >= (1ULL << 53) || e10 < -22 || e10 > 22) goto slow;
  {
    double d = (double)m;
    d = e10 >= 0 ? d * p10[e10] : d / p10[-e10];
    if (!(d >= 1.2e-38 && d <= 3.4e38)) goto slow;          /* keep clear of the float range limits */
    unsigned long long bits;
    memcpy(&bits, &d, sizeof bits);
    if ((bits & 0x1FFFFFFFULL) == 0x10000000ULL) goto slow;  /* exactly on a float tie: let strtof decide */
    *out = (float)(neg ? -d : d);
    return 1;
  }
slow:
  return sscanf(s, "%f", out) > 0;
}

/* what may follow the numbers of a row: labels, weight=N, fixed=X,Y (datafile.c:705-735) */
static int row_tokens(struct entries *e, long r, char *tok, char **save, struct fixpoint *fix, int *any_weight,
                      int *any_fixed, int labels_needed, long lineno, const char *name)
{
  struct data_entry *d = &e->rows[r];
  int label_found = 0;
  for (; tok; tok = strtok_r(NULL, " \r\t", save)) {
    if (strncmp(tok, "weight=", 7) == 0) { d->weight = (short)atoi(tok + 7); *any_weight = 1; }
    else if (strncmp(tok, "fixed=", 6) == 0) {
      char *comma = strchr(tok, ',');
      if (!comma) { fprintf(stderr, "bad fixed point, line %ld of file %s\n", lineno, name); return 1; }
      fix->xfix = (short)atoi(tok + 6);
      fix->yfix = (short)atoi(comma + 1);
      *any_fixed = 1;
    } else {
      add_entry_label(e, r, find_conv_to_ind(tok));
      label_found++;
    }
  }
  if (labels_needed && !label_found) {
    fprintf(stderr, "Required label missing on line %ld of file %s\n", lineno, name);
    return 1;
  }
  return 0;
}

static int parse_header(struct entries *e, const char *line, const char *name)
{
  int dim = 0;
  if (sscanf(line, "%d", &dim) <= 0 || dim <= 0) {
    fprintf(stderr, "Can't read dimension parameter in file %s", name);
    return 0;
  }
  char *save, *dup = strdup(line);
  strtok_r(dup, " ", &save);
  char *t = strtok_r(NULL, " ", &save);
  char *xs = strtok_r(NULL, " ", &save), *ys = strtok_r(NULL, " ", &save), *ns = strtok_r(NULL, " ", &save);
  e->dimension = (short)dim;
  e->topol = (short)id_of(topol_names, 5, t);
  e->xdim = xs ? (short)atoi(xs) : 0;
  e->ydim = ys ? (short)atoi(ys) : 0;
  e->neigh = (short)id_of(neigh_names, 3, ns);
  free(dup);
  return dim;
}

/* dense side arrays + row views once all rows are in e->points (maskrows / fixtmp may be NULL) */
static void finish_entries(struct entries *e, char **maskrows, struct fixpoint *fixtmp, int any_fixed, int any_weight)
{
  long n = e->num_entries;
  int dim = e->dimension, any_mask = 0;
  for (long r = 0; maskrows && r < n; r++) any_mask |= maskrows[r] != NULL;
  if (any_mask) e->masks = calloc((size_t)n * dim + 1, 1);
  if (any_fixed) e->fixed_xy = malloc(sizeof(short) * 2 * (n + 1));
  if (any_weight) e->weights = malloc(sizeof(short) * (n + 1));
  for (long r = 0; r < n; r++) {
    struct data_entry *d = &e->rows[r];
    d->points = e->points + r * dim;
    if (any_mask && maskrows[r]) { memcpy(e->masks + r * dim, maskrows[r], dim); d->mask = e->masks + r * dim; }
    if (maskrows) free(maskrows[r]);
    if (any_fixed) {
      e->fixed_xy[2 * r] = fixtmp[r].xfix; e->fixed_xy[2 * r + 1] = fixtmp[r].yfix;
      if (fixtmp[r].xfix >= 0) d->fixed = (struct fixpoint *)(e->fixed_xy + 2 * r);
    }
    if (any_weight) e->weights[r] = d->weight;
  }
}

/* ---- raw fp32 side format (the text parser is the wall once the kernels are fast) ----
 *   line 1   "#!somf32 <rows> <flags>"         flags bit 0: a text section follows the numbers
 *   line 2   the .dat header line              "<dim> [topol [xdim ydim neigh]]"  (datafile.c:396-415)
 *   payload  rows * dim little-endian float32  NaN = masked component (the 'x' of the text format)
 *   text     (flag bit 0) one line per row with what follows the numbers in a .dat row: labels, weight=, fixed=
 * Read with one fread straight into the dense array; rows whose components are all masked are dropped as in
 * the text reader.  `datconv` converts both ways. */
static struct entries *read_f32(FILE *fp, const char *first_line, const char *name, int labels_needed, int skip_empty)
{
  long n = 0;
  int flags = 0;
  if (sscanf(first_line, "#!somf32 %ld %d", &n, &flags) < 1 || n < 0) { fprintf(stderr, "bad somf32 header in file %s\n", name); return NULL; }
  struct entries *e = calloc(1, sizeof *e);
  e->labels_needed = labels_needed;
  char *line = NULL;
  size_t cap = 0;
  struct fixpoint *fixtmp = NULL;
  char **maskrows = NULL;
  int any_fixed = 0, any_weight = 0, dim;
  if (getline(&line, &cap, fp) < 0) goto fail;
  { size_t L = strlen(line); while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0; }
  if (!(dim = parse_header(e, line, name))) goto fail;
  e->points = malloc(sizeof(float) * (size_t)(n ? n : 1) * dim);
  e->rows = calloc((size_t)(n ? n : 1), sizeof(struct data_entry));
  if (fread(e->points, sizeof(float) * dim, (size_t)n, fp) != (size_t)n) { fprintf(stderr, "file %s is shorter than its header says\n", name); goto fail; }
  maskrows = calloc((size_t)(n ? n : 1), sizeof(char *));
  fixtmp = malloc(sizeof(struct fixpoint) * (size_t)(n ? n : 1));
  long kept = 0;
  for (long r = 0; r < n; r++) {                    /* NaN -> mask; drop empty rows; compact in place */
    float *p = e->points + r * dim;
    char *mask = NULL;
    int maskcnt = 0;
    for (int i = 0; i < dim; i++)
      if (p[i] != p[i]) { if (!mask) mask = calloc(dim, 1); mask[i] = 1; maskcnt++; p[i] = 0.0f; }
    char *tokline = NULL, *save = NULL, *tok = NULL;
    if (flags & 1) {
      if (getline(&line, &cap, fp) < 0) { fprintf(stderr, "file %s: text section ends at row %ld\n", name, r); free(mask); goto fail; }
      size_t L = strlen(line);
      if (L && line[L - 1] == '\n') line[--L] = 0;
      tokline = line;
    }
    if (maskcnt == dim && skip_empty) { free(mask); continue; }
    if (kept != r) memmove(e->points + kept * dim, p, sizeof(float) * dim);
    maskrows[kept] = mask;
    fixtmp[kept].xfix = fixtmp[kept].yfix = -1;
    e->num_entries = kept + 1;
    if (tokline) tok = strtok_r(tokline, " \r\t", &save);
    if (row_tokens(e, kept, tok, &save, &fixtmp[kept], &any_weight, &any_fixed, labels_needed, r + 3, name)) goto fail;
    kept++;
  }
  e->num_entries = kept;
  finish_entries(e, maskrows, fixtmp, any_fixed, any_weight);
  free(maskrows); free(fixtmp); free(line);
  return e;
fail:
  free(maskrows); free(fixtmp); free(line);
  close_entries(e);
  return NULL;
}

int save_entries_f32(struct entries *c, const char *name)
{
  int is_pipe, any_text = 0;
  for (long r = 0; r < c->num_entries && !any_text; r++)
    any_text = c->rows[r].num_labs > 0 || c->rows[r].weight != 0 || c->rows[r].fixed != NULL;
  FILE *fp = open_text(name, "w", &is_pipe);
  if (!fp) { fprintf(stderr, "Can't open file %s for writing\n", name); return 1; }
  fprintf(fp, "#!somf32 %ld %d\n%d", c->num_entries, any_text, c->dimension);
  if (c->topol > TOPOL_DATA) {
    fprintf(fp, " %s", topol_names[c->topol]);
    if (c->topol > TOPOL_LVQ) fprintf(fp, " %d %d %s", c->xdim, c->ydim, neigh_names[c->neigh] ? neigh_names[c->neigh] : "");
  }
  fputc('\n', fp);
  const int dim = c->dimension;
  float *tmp = malloc(sizeof(float) * dim);
  for (long r = 0; r < c->num_entries; r++) {
    const struct data_entry *d = &c->rows[r];
    if (d->mask) {
      for (int i = 0; i < dim; i++) tmp[i] = d->mask[i] ? __builtin_nanf("") : d->points[i];
      fwrite(tmp, sizeof(float), dim, fp);
    } else fwrite(d->points, sizeof(float), dim, fp);
  }
  free(tmp);
  for (long r = 0; any_text && r < c->num_entries; r++) {
    const struct data_entry *d = &c->rows[r];
    for (int k = 0; k < d->num_labs; k++) fprintf(fp, "%s ", find_conv_to_lab(d->labels[k]));
    if (d->weight) fprintf(fp, "weight=%d ", d->weight);
    if (d->fixed) fprintf(fp, "fixed=%d,%d ", d->fixed->xfix, d->fixed->yfix);
    fputc('\n', fp);
  }
  int bad = ferror(fp);
  close_text(fp, is_pipe);
  return bad;
}

/* ---- seeded generator as a data source:  -din gen:k=256,dim=512,n=100000,seed=3456[,labels=1] ----
 * The Gaussian-mixture stream of SURVEY 8(d), counter-based so that any row can be produced anywhere (host here,
 * k_gen_mixture on the device: same bits, tests/test_gpu_parity.py):  splitmix64(seed ^ counter) words; a centre
 * component is 4 z, a sample is centre[k(row)] + z with k(row) = word(seed_assign ^ row) mod K; z is the classic
 * sum of twelve uniforms minus six, here twelve 16-bit fields of three words: integer arithmetic and one exact
 * division by 65536, so host and device cannot differ (a Box-Muller z would depend on each side's log and cos).
 * labels=1 attaches the mixture id ("c<k>") as the row's label. */
uint64_t pak_splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
float pak_gen_z(uint64_t seed, uint64_t counter)
{
  int32_t sum = 0;
  for (int w = 0; w < 3; w++) {
    uint64_t v = pak_splitmix64(seed ^ (3 * counter + w));
    sum += (int32_t)(v & 0xFFFF) + (int32_t)((v >> 16) & 0xFFFF) + (int32_t)((v >> 32) & 0xFFFF) + (int32_t)(v >> 48);
  }
  return (float)(sum - 6 * 65535) / 65536.0f;      /* |sum - 393210| < 2^24: exact */
}
void pak_gen_row(uint64_t seed, int k_centres, int dim, long row, float *out, int *centre)
{
  const uint64_t seed_c = seed ^ 0xC3A5C85C97CB3127ULL, seed_a = seed ^ 0xB492B66FBE98F273ULL;
  const int k = (int)(pak_splitmix64(seed_a ^ (uint64_t)row) % (uint64_t)k_centres);
  for (int i = 0; i < dim; i++) {
    const float mu = 4.0f * pak_gen_z(seed_c, (uint64_t)k * dim + i);
    out[i] = mu + pak_gen_z(seed, (uint64_t)row * dim + i);
  }
  if (centre) *centre = k;
}
int pak_parse_gen(const char *spec, long *n, int *dim, int *k, uint64_t *seed, int *labels)
{
  if (strncmp(spec, "gen:", 4) != 0) return 0;
  *n = 0; *dim = 0; *k = 16; *seed = 1234; *labels = 0;
  char *dup = strdup(spec + 4), *save, *tok;
  for (tok = strtok_r(dup, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
    if (sscanf(tok, "n=%ld", n) == 1 || sscanf(tok, "dim=%d", dim) == 1 || sscanf(tok, "k=%d", k) == 1 ||
        sscanf(tok, "labels=%d", labels) == 1) continue;
    unsigned long long sv;
    if (sscanf(tok, "seed=%llu", &sv) == 1) { *seed = sv; continue; }
    fprintf(stderr, "gen: unknown field '%s' (n=, dim=, k=, seed=, labels=)\n", tok);
    free(dup);
    return -1;
  }
  free(dup);
  if (*n <= 0 || *dim <= 0 || *dim > 32767 || *k <= 0) { fprintf(stderr, "gen: needs n=, dim= (and k > 0)\n"); return -1; }
  return 1;
}
int pak_gen_virtual_ok = 0;

/* host rows of a virtual source (same stream, same bits as the device's) */
int pak_materialize(struct entries *e)
{
  if (!e || !e->is_virtual) return 0;
  const long n = e->num_entries;
  const int dim = e->dimension;
  e->points = malloc(sizeof(float) * (size_t)n * dim);
  e->rows = calloc((size_t)n, sizeof(struct data_entry));
  if (!e->points || !e->rows) { fprintf(stderr, "gen: out of memory for %ld x %d host rows\n", n, dim); return 1; }
#pragma omp parallel for schedule(static)
  for (long r = 0; r < n; r++) pak_gen_row(e->gen_seed, e->gen_k, dim, r, e->points + r * dim, NULL);
  e->is_virtual = 0;
  finish_entries(e, NULL, NULL, 0, 0);
  return 0;
}

static struct entries *gen_entries(const char *spec)
{
  long n; int dim, k, labels; uint64_t seed;
  if (pak_parse_gen(spec, &n, &dim, &k, &seed, &labels) <= 0) return NULL;
  struct entries *e = calloc(1, sizeof *e);
  e->dimension = (short)dim;
  e->num_entries = n;
  if (pak_gen_virtual_ok && !labels) {                 /* kept as a specification: the engine generates it in HBM */
    e->is_virtual = 1; e->gen_seed = seed; e->gen_k = k;
    return e;
  }
  e->points = malloc(sizeof(float) * (size_t)n * dim);
  e->rows = calloc((size_t)n, sizeof(struct data_entry));
  int *centre = malloc(sizeof(int) * (size_t)n);
  /* counter-based: every row is independent, so the rows are made in parallel (the label table is not thread-safe:
   * labels are attached afterwards, in row order, which is also the order the text reader would meet them in) */
#pragma omp parallel for schedule(static)
  for (long r = 0; r < n; r++) pak_gen_row(seed, k, dim, r, e->points + r * dim, &centre[r]);
  for (long r = 0; labels && r < n; r++) {
    char nm[32];
    snprintf(nm, sizeof nm, "c%d", centre[r]);
    add_entry_label(e, r, find_conv_to_ind(nm));
  }
  free(centre);
  finish_entries(e, NULL, NULL, 0, 0);
  return e;
}

/* open_entries + read_entries (datafile.c:191,237) for a whole file.  Header: first
 * non-comment line "<dim> [topol [xdim ydim neigh]]" (datafile.c:112-145).  Rows: <dim>
 * numbers or the mask string, then labels / weight=N / fixed=X,Y (datafile.c:552-748);
 * '#' lines and blank lines are skipped; rows with every component masked are dropped
 * when skip_empty (datafile.c:677-686). */
struct entries *open_entries(const char *name, int labels_needed, int skip_empty)
{
  int is_pipe;
  if (strncmp(name, "gen:", 4) == 0) return gen_entries(name);
  FILE *fp = open_text(name, "r", &is_pipe);
  if (!fp) { fprintf(stderr, "Can't open file %s", name); return NULL; }
  struct entries *e = calloc(1, sizeof *e);
  e->labels_needed = labels_needed;
  char *line = NULL;
  size_t cap = 0;
  long lineno = 0, nalloc = 0;
  int have_header = 0, dim = 0;
  struct fixpoint *fixtmp = NULL;
  int any_fixed = 0, any_weight = 0;
  char **maskrows = NULL;

  while (getline(&line, &cap, fp) >= 0) {
    lineno++;
    size_t L = strlen(line);
    if (L && line[L - 1] == '\n') line[--L] = 0;
    if (lineno == 1 && strncmp(line, "#!somf32", 8) == 0) {      /* the raw fp32 side format */
      struct entries *b = read_f32(fp, line, name, labels_needed, skip_empty);
      free(line); free(e);
      close_text(fp, is_pipe);
      return b;
    }
    if (line[0] == '#') continue;
    if (!have_header) {
      if (!(dim = parse_header(e, line, name))) goto fail;
      have_header = 1;
      continue;
    }
    char *save;
    char *tok = strtok_r(line, " \r\t", &save);
    if (!tok) continue;                          /* empty line */
    if (e->num_entries == nalloc) {
      nalloc = nalloc ? 2 * nalloc : 1024;
      e->points = realloc(e->points, sizeof(float) * nalloc * dim);
      e->rows = realloc(e->rows, sizeof(struct data_entry) * nalloc);
      maskrows = realloc(maskrows, sizeof(char *) * nalloc);
      fixtmp = realloc(fixtmp, sizeof(struct fixpoint) * nalloc);
    }
    long r = e->num_entries;
    float *p = e->points + r * dim;
    char *mask = NULL;
    int maskcnt = 0;
    for (int i = 0; i < dim; i++) {
      if (i > 0) tok = strtok_r(NULL, " \r\t", &save);
      if (!tok) {
        fprintf(stderr, "load_entry: can't read entry in file %s on line %ld, component %d\n", name, lineno, i);
        goto fail;
      }
      if (strcmp(tok, masked_string) == 0) {
        if (!mask) mask = calloc(dim, 1);
        mask[i] = 1; maskcnt++; p[i] = 0.0f;
      } else if (!pak_parse_float(tok, &p[i])) {
        fprintf(stderr, "load_entry: can't read entry in file %s on line %ld, component %d\n", name, lineno, i);
        goto fail;
      }
    }
    if (maskcnt == dim && skip_empty) { free(mask); continue; }
    struct data_entry *d = &e->rows[r];
    memset(d, 0, sizeof *d);
    maskrows[r] = mask;
    fixtmp[r].xfix = fixtmp[r].yfix = -1;
    e->num_entries++;
    tok = strtok_r(NULL, " \r\t", &save);
    if (row_tokens(e, r, tok, &save, &fixtmp[r], &any_weight, &any_fixed, labels_needed, lineno, name)) goto fail;
  }
  if (!have_header) { fprintf(stderr, "Can't read file %s", name); goto fail; }
  finish_entries(e, maskrows, fixtmp, any_fixed, any_weight);
  free(maskrows); free(fixtmp); free(line);
  close_text(fp, is_pipe);
  return e;
fail:
  free(line);
  close_text(fp, is_pipe);
  return NULL;
}

void close_entries(struct entries *e)
{
  if (!e) return;
  if (e->userdata) somhip_codebook_destroy(e->userdata);     /* the per-sample surface's mirror (an orphan if the engine went first) */
  for (long r = 0; e->rows && r < e->num_entries; r++) free(e->rows[r].labels);   /* (a virtual gen: source has no rows) */
  free(e->rows); free(e->points); free(e->masks); free(e->fixed_xy); free(e->weights);
  free(e);
}

/* write_header datafile.c:396-415, write_entry :420-447: "%g " per value, "%s " per label */
static void write_rows(FILE *fp, struct entries *c, const char *comments)
{
  fprintf(fp, "%d", c->dimension);
  if (c->topol > TOPOL_DATA) {
    fprintf(fp, " %s", topol_names[c->topol]);
    if (c->topol > TOPOL_LVQ) fprintf(fp, " %d %d %s", c->xdim, c->ydim, neigh_names[c->neigh] ? neigh_names[c->neigh] : "");
  }
  fprintf(fp, "\n");
  if (comments) fputs(comments, fp);
  for (long r = 0; r < c->num_entries; r++) {
    struct data_entry *d = &c->rows[r];
    for (int i = 0; i < c->dimension; i++)
      if (d->mask && d->mask[i]) fprintf(fp, "%s ", masked_string);
      else fprintf(fp, "%g ", d->points[i]);
    for (int k = 0; k < d->num_labs; k++) {
      if (d->labels[k] == LABEL_EMPTY) break;
      fprintf(fp, "%s ", find_conv_to_lab(d->labels[k]));
    }
    fprintf(fp, "\n");
  }
}
int save_entries_wcomments(struct entries *codes, const char *name, const char *comments)
{
  int is_pipe;
  size_t nl = strlen(name);
  if (nl > 4 && strcmp(name + nl - 4, ".f32") == 0)      /* a name ending in .f32 asks for the raw fp32 side format */
    return save_entries_f32(codes, name);                /* (a 256x256x512 codebook is 400 MB of "%g" text otherwise)  */
  FILE *fp = open_text(name, "w", &is_pipe);
  if (!fp) { fprintf(stderr, "save_entries: Can't open file '%s'\n", name); return 1; }
  write_rows(fp, codes, comments);
  close_text(fp, is_pipe);
  return 0;
}

/* OLVQ1 learning-rate files, datafile.c:1030-1110: "<name up to the first '.'>.lra", one
 * "%g" per line */
static void lra_name(char *out, size_t n, const char *file)
{
  snprintf(out, n - 4, "%s", file);
  char *dot = strchr(out, '.');
  if (dot) *dot = 0;
  strcat(out, ".lra");
}
int alpha_read(float *alpha, long noc, const char *infile)
{
  char nm[2048];
  lra_name(nm, sizeof nm, infile);
  FILE *fp = fopen(nm, "r");
  if (!fp) { ifverbose(1) fprintf(stderr, "Can't open alpha file %s", nm); return 0; }
  for (long i = 0; i < noc; i++)
    if (fscanf(fp, "%g\n", &alpha[i]) < 0) { fclose(fp); return 0; }
  fclose(fp);
  return 1;
}
int alpha_write(float *alpha, long noc, const char *outfile)
{
  char nm[2048];
  lra_name(nm, sizeof nm, outfile);
  FILE *fp = fopen(nm, "w+");
  if (!fp) { fprintf(stderr, "Can't open alpha file %s for writing", nm); return 0; }
  for (long i = 0; i < noc; i++) fprintf(fp, "%g\n", alpha[i]);
  fclose(fp);
  return 0;
}
void invalidate_alphafile(const char *outfile)
{
  char nm[2048];
  lra_name(nm, sizeof nm, outfile);
  FILE *fp = fopen(nm, "r");
  if (fp) {
    ifverbose(1) fprintf(stdout, "Removing the learning rate file %s\n", nm);
    fclose(fp);
    if (remove(nm)) fprintf(stderr, "Can not remove %s", nm);
  }
}

/* ------------------------------------------------------------------ RNG, shuffle */
static unsigned long rnd_next = 1;
void init_random(int seed) { rnd_next = seed ? (unsigned long)seed : (unsigned long)(int)time(NULL); }
long orand(void) { rnd_next = (rnd_next * 23UL) % 100000001UL; return (long)(int)(rnd_next % 32767UL); }

/* datafile.c:1152-1188: for i in order, swap slot i with slot orand() % n */
void randomize_entry_order(struct entries *e)
{
  long n = e->num_entries;
  int dim = e->dimension;
  if (n <= 0) return;
  long *perm = malloc(sizeof(long) * n);
  for (long i = 0; i < n; i++) perm[i] = i;
  for (long i = 0; i < n; i++) {
    long j = orand() % n, t = perm[i];
    perm[i] = perm[j]; perm[j] = t;
  }
  float *pts = malloc(sizeof(float) * (size_t)n * dim);
  struct data_entry *rows = malloc(sizeof(struct data_entry) * n);
  char *masks = e->masks ? calloc((size_t)n * dim + 1, 1) : NULL;
  short *fx = e->fixed_xy ? malloc(sizeof(short) * 2 * (n + 1)) : NULL;
  short *wt = e->weights ? malloc(sizeof(short) * (n + 1)) : NULL;
  for (long i = 0; i < n; i++) {
    long s = perm[i];
    memcpy(pts + i * dim, e->points + s * dim, sizeof(float) * dim);
    rows[i] = e->rows[s];
    rows[i].points = pts + i * dim;
    if (masks) {
      memcpy(masks + i * dim, e->masks + s * dim, dim);
      rows[i].mask = e->rows[s].mask ? masks + i * dim : NULL;
    }
    if (fx) {
      fx[2 * i] = e->fixed_xy[2 * s]; fx[2 * i + 1] = e->fixed_xy[2 * s + 1];
      rows[i].fixed = fx[2 * i] >= 0 ? (struct fixpoint *)(fx + 2 * i) : NULL;
    }
    if (wt) wt[i] = e->weights[s];
  }
  free(e->points); free(e->rows); free(e->masks); free(e->fixed_xy); free(e->weights); free(perm);
  e->points = pts; e->rows = rows; e->masks = masks; e->fixed_xy = fx; e->weights = wt;
}

/* ------------------------------------------------------------------ schedules (host scalars) */
float linear_alpha(long iter, long length, float alpha)      /* lvq_pak.c:903-906 */
{
  return alpha * (float)(length - iter) / (float)length;
}
float inverse_t_alpha(long iter, long length, float alpha)   /* lvq_pak.c:914-921 */
{
  float c = (float)length / 100.0f;
  return alpha * c / (c + (float)iter);
}
ALPHA_FUNC *alpha_func_by_name(const char *name, short *id)
{
  if (!name || strcasecmp(name, "linear") == 0) { *id = ALPHA_LINEAR; return linear_alpha; }
  if (strcasecmp(name, "inverse_t") == 0) { *id = ALPHA_INVERSE_T; return inverse_t_alpha; }
  *id = ALPHA_UNKNOWN;
  return NULL;
}

/* ------------------------------------------------------------------ shared tool front end */
int pak_open_inputs(const char *din, int data_labels, const char *data_fail_fmt, const char *cin, int code_labels,
                    const char *code_fail_fmt, int need_map, struct pak_inputs *io)
{
  io->data = io->codes = NULL;
  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", din);
  if (!(io->data = open_entries(din, data_labels, 1))) { fprintf(stderr, data_fail_fmt, din); return 1; }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", cin);
  if (!(io->codes = open_entries(cin, code_labels, 1))) { fprintf(stderr, code_fail_fmt, cin); goto bad; }
  if (need_map && io->codes->topol < TOPOL_HEXA) { fprintf(stderr, "File %s is not a map file\n", cin); goto bad; }
  if (io->data->dimension != io->codes->dimension) {
    fprintf(stderr, need_map == 2 ? "Data and codebook vectors have different dimensions (%d != %d)"
                                  : "Data and codebook vectors have different dimensions",
            io->data->dimension, io->codes->dimension);
    goto bad;
  }
  return 0;
bad:
  close_entries(io->data); close_entries(io->codes);
  io->data = io->codes = NULL;
  return 1;
}

void pak_train_cli(int argc, char **argv, struct pak_train_cli *o)
{
  memset(o, 0, sizeof *o);
  o->din = extract_parameter(argc, argv, "-din", ALWAYS);
  o->cin = extract_parameter(argc, argv, "-cin", ALWAYS);
  o->cout = extract_parameter(argc, argv, "-cout", ALWAYS);
  o->length = oatoi(extract_parameter(argc, argv, "-rlen", ALWAYS), 1);
  o->rand_s = extract_parameter(argc, argv, "-rand", OPTION);
  o->buffer = oatoi(extract_parameter(argc, argv, "-buffer", OPTION), 0);
  o->alpha_s = extract_parameter(argc, argv, "-alpha_type", OPTION);
  o->funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  o->snap.filename = extract_parameter(argc, argv, "-snapfile", OPTION);
  o->snap.interval = oatoi(extract_parameter(argc, argv, "-snapinterval", OPTION), 0);
  o->want_snapshots = o->snap.interval != 0;
  if (o->want_snapshots && !o->snap.filename) {
    o->snap.filename = o->cout;
    fprintf(stderr, "snapshot file not specified, using '%s'", o->snap.filename);
  }
}

void pak_apply_rand(struct entries *data, const char *rand_s, long buffer)
{
  init_random((int)oatoi(rand_s, 0));
  if (!rand_s) return;
  if (pak_materialize(data)) exit(1);                  /* a shuffle needs the rows on the host */
  if (buffer > 0 && buffer < data->num_entries) { data->buffer = buffer; data->random_order = 1; }   /* reshuffled per buffer */
  else randomize_entry_order(data);                                                               /* once, at load */
}

/* ------------------------------------------------------------------ the HIP back end */
static somhip_engine *g_engine = NULL;
static int g_device = 0;                               /* a rank of a multi-GPU run sets this before its first engine() */

static somhip_engine *engine(void)
{
  if (!g_engine && somhip_engine_create(g_device, &g_engine)) {
    fprintf(stderr, "%s\n", somhip_last_error());
    return NULL;
  }
  return g_engine;
}
void pak_shutdown(void) { if (g_engine) { somhip_engine_destroy(g_engine); g_engine = NULL; } }

static int32_t *first_labels(struct entries *e)
{
  int32_t *l = malloc(sizeof(int32_t) * (e->num_entries + 1));
  for (long r = 0; r < e->num_entries; r++) l[r] = get_entry_label(&e->rows[r]);
  return l;
}

static somhip_codebook *mirror_codes(struct entries *codes, int with_labels)
{
  somhip_engine *en = engine();
  if (!en) return NULL;
  somhip_codebook *cb = NULL;
  int32_t *lab = with_labels ? first_labels(codes) : NULL;
  int rc = somhip_codebook_create(en, codes->points, lab, codes->num_entries, codes->dimension, codes->topol,
                                  codes->neigh, codes->xdim, codes->ydim, 0, codes->num_entries, &cb);
  free(lab);
  if (rc) { fprintf(stderr, "%s\n", somhip_last_error()); return NULL; }
  return cb;
}
static somhip_dataset *mirror_data(struct entries *data, int with_labels)
{
  somhip_engine *en = engine();
  if (!en) return NULL;
  somhip_dataset *ds = NULL;
  if (data->is_virtual) {
    if (with_labels && pak_materialize(data)) return NULL;
    if (data->is_virtual) {
      if (somhip_dataset_generate(en, data->gen_seed, data->gen_k, data->dimension, 0, data->num_entries, NULL, &ds)) {
        fprintf(stderr, "%s\n", somhip_last_error());
        return NULL;
      }
      return ds;
    }
  }
  int32_t *lab = with_labels ? first_labels(data) : NULL;
  int rc = somhip_dataset_create(en, data->points, data->num_entries, data->dimension,
                                 (const uint8_t *)data->masks, lab, data->weights, data->fixed_xy, &ds);
  free(lab);
  if (rc) { fprintf(stderr, "%s\n", somhip_last_error()); return NULL; }
  return ds;
}

/* The per-sample surface of the "hip" row (lvq_pak.h:131-148).  Tools should use the epoch-level functions; these
 * keep per-sample callers of the registry working (balance.c:56, som_rout.c:741,785 dereference dist; the
 * reference's own loops call winner and vector_adapt once per sample).
 *
 * dist / vector_adapt act on ONE host row each -- the host list owns the rows (SURVEY 8b "Ownership") and a
 * PCIe round trip per row would cost a thousand times the arithmetic -- so they are the reference's own
 * expressions on the host rows: vector_dist_euc (lvq_pak.c:291-316) and adapt_vector (lvq_pak.c:339-351),
 * compiled without contraction.  A row written by vector_adapt makes every device mirror stale: the generation
 * counter below makes the next winner call re-upload its codebook. */
static unsigned long host_rows_generation = 0;

static float hip_row_dist(struct data_entry *v1, struct data_entry *v2, int dim)
{
  float sum = 0.0f;
  int masked = 0;
  for (int i = 0; i < dim; i++) {
    if ((v1->mask && v1->mask[i]) || (v2->mask && v2->mask[i])) { masked++; continue; }
    float t = v1->points[i] - v2->points[i];
    sum += t * t;
  }
  if (masked == dim) return -1.0f;                    /* nothing to compare (lvq_pak.c:312-313) */
  return (float)sqrt((double)sum);
}

static void hip_row_adapt(struct data_entry *code, struct data_entry *sample, int dim, float alpha)
{
  for (int i = 0; i < dim; i++) {
    if (sample->mask && sample->mask[i]) continue;    /* only the sample's mask counts (lvq_pak.c:345-346) */
    code->points[i] += alpha * (sample->points[i] - code->points[i]);
  }
  host_rows_generation++;
}

static int hip_find_winner(struct entries *codes, struct data_entry *sample, struct winner_info *w, int knn)
{
  somhip_codebook *cb = codes->userdata;
  if (!cb) { cb = codes->userdata = mirror_codes(codes, 0); codes->mirror_generation = host_rows_generation; }
  if (!cb) return 0;
  if (codes->mirror_generation != host_rows_generation) {       /* rows were adapted on the host since the upload */
    if (somhip_codebook_upload(cb, codes->points)) { fprintf(stderr, "%s\n", somhip_last_error()); return 0; }
    codes->mirror_generation = host_rows_generation;
  }
  somhip_dataset *ds = NULL;
  int32_t idx[8], ret = 0;
  float diff[8];
  if (knn < 1 || knn > 8) return 0;
  if (somhip_dataset_create(engine(), sample->points, 1, codes->dimension, (const uint8_t *)sample->mask,
                            NULL, NULL, NULL, &ds)) return 0;
  int rc = somhip_find_winners(cb, ds, 0, 1, knn, knn > 1 ? SOMHIP_TIE_KNN : SOMHIP_TIE_FIRST, idx, diff, &ret);
  somhip_dataset_destroy(ds);
  if (rc) { fprintf(stderr, "%s\n", somhip_last_error()); return 0; }
  for (int k = 0; k < knn; k++) {
    w[k].index = idx[k];
    w[k].winner = idx[k] >= 0 ? &codes->rows[idx[k]] : NULL;
    w[k].diff = diff[k];
  }
  return ret;
}

/* the registry (datafile.c:1207-1243).  There is one row, "hip"; the reference's name
 * "default" is accepted as an alias so existing command lines run unchanged.  Unknown names
 * warn and fall back exactly as the reference does. */
static struct vec_functions { const char *name; DIST_FUNCTION *dist; VECTOR_ADAPT *vector_adapt; WINNER_FUNCTION *winner; }
vec_funcs[] = { {"hip", hip_row_dist, hip_row_adapt, hip_find_winner}, {"default", hip_row_dist, hip_row_adapt, hip_find_winner},
                {NULL, NULL, NULL, NULL} };

int set_teach_params(struct teach_params *p, struct entries *codes, struct entries *data, const char *funcname)
{
  struct vec_functions *v = vec_funcs;
  if (funcname)
    for (; v->name; v++) if (strcasecmp(v->name, funcname) == 0) break;
  if (funcname && !v->name) {
    fprintf(stderr, "functions for '%s' not found, using defaults\n", funcname);
    v = vec_funcs;
  }
  p->topol = codes->topol; p->neigh = codes->neigh;
  p->mapdist = NULL; p->neigh_adapt = NULL;
  p->dist = v->dist; p->vector_adapt = v->vector_adapt; p->winner = v->winner;
  p->codes = codes;
  if (data) p->data = data;
  p->snapshot = NULL;
  p->batch = 1;
  return 0;
}
int set_som_params(struct teach_params *p)                   /* som_rout.c:936-947 */
{
  if (p->topol != TOPOL_HEXA && p->topol != TOPOL_RECT) return 1;
  if (p->neigh != NEIGH_BUBBLE && p->neigh != NEIGH_GAUSSIAN) return 1;
  return 0;
}

static int save_snapshot(struct teach_params *teach, long iter)   /* lvq_pak.c:665-774, synchronous form */
{
  char filename[1024], comment[128];
  snprintf(filename, sizeof filename, teach->snapshot->filename, iter);
  snprintf(comment, sizeof comment, "#SNAPSHOT FILE\n#iterations: %ld/%ld\n", iter, teach->length);
  teach->snapshot->counter++;
  return save_entries_wcomments(teach->codes, filename, comment);
}

/* iterations are run in segments that end where the reference would save a snapshot
 * (after iteration le, when le % interval == 0 && le > 0: som_rout.c:650, lvq_rout.c:559) */
static long segment_end(struct teach_params *t, long start)
{
  if (!t->snapshot || t->snapshot->interval <= 0) return t->length;
  const long iv = t->snapshot->interval;
  long next = (start + iv - 1) / iv * iv;             /* smallest le >= start with le % iv == 0 ... */
  if (next == 0) next = iv;                           /* ... and le > 0 (som_rout.c:650) */
  long end = next + 1;
  return end < t->length ? end : t->length;
}

/* The order in which a training run sees the data (datafile.c:237-344, 754-830).  Whole file in
 * memory: the rows as they are (already shuffled once if -rand), cyclically.  -buffer N together
 * with -rand: rows [0,N), [N,2N), ... of the FILE, each buffer shuffled with the running orand()
 * when it is loaded, the file rewound after the last buffer and every run starting at the top.
 * Each buffer becomes a device data set of its own. */
struct feed { struct entries *data, *sub; somhip_dataset *ds; long pos, left, first; int per_buffer, with_labels; };

static int feed_open(struct feed *f, struct entries *data, int with_labels)
{
  memset(f, 0, sizeof *f);
  f->data = data; f->with_labels = with_labels;
  f->per_buffer = data->random_order && data->buffer > 0 && data->buffer < data->num_entries;
  if (!f->per_buffer) { f->ds = mirror_data(data, with_labels); f->left = -1; return f->ds ? 0 : 1; }
  return 0;
}
/* make sure at least one row is available; returns how many consecutive rows can be taken now
 * (*first = index of the next one inside the current device data set), 0 on failure */
static long feed_avail(struct feed *f, long iter, long *first)
{
  if (!f->per_buffer) { *first = iter % f->data->num_entries; return LONG_MAX; }
  if (f->left == 0 || !f->ds) {
    if (f->ds) { somhip_dataset_destroy(f->ds); f->ds = NULL; }
    if (f->sub) { close_entries(f->sub); f->sub = NULL; }
    long n = f->data->num_entries, nb = f->data->buffer < n - f->pos ? f->data->buffer : n - f->pos;
    long *idx = malloc(sizeof(long) * nb);
    for (long i = 0; i < nb; i++) idx[i] = f->pos + i;
    for (long i = 0; i < nb; i++) { long j = orand() % nb, t = idx[i]; idx[i] = idx[j]; idx[j] = t; }   /* datafile.c:1171-1177 */
    f->sub = pick_rows(f->data, idx, nb);
    free(idx);
    f->ds = mirror_data(f->sub, f->with_labels);
    if (!f->ds) return 0;
    f->left = nb; f->first = 0;
    f->pos = f->pos + nb >= n ? 0 : f->pos + nb;
  }
  *first = f->first;
  return f->left;
}
static void feed_took(struct feed *f, long count) { if (f->per_buffer) { f->left -= count; f->first += count; } }
static void feed_close(struct feed *f)
{
  if (f->ds) somhip_dataset_destroy(f->ds);
  if (f->sub) close_entries(f->sub);
}

struct entries *som_training(struct teach_params *teach)     /* som_rout.c:556-671 */
{
  struct entries *codes = teach->codes, *data = teach->data;
  if (set_som_params(teach)) { fprintf(stderr, "som_training: can't set SOM parameters\n"); return NULL; }
  if (!data || data->num_entries <= 0) { fprintf(stderr, "som_training: can't get data\n"); return NULL; }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "code dimension (%d) != data dimension (%d)\n", codes->dimension, data->dimension);
    return NULL;
  }
  somhip_codebook *cb = mirror_codes(codes, 0);
  struct feed fd;
  struct entries *ret = NULL;
  if (feed_open(&fd, data, 0) || !cb) goto done;
  for (long start = 0; start < teach->length;) {
    long first, avail = feed_avail(&fd, start, &first);
    if (avail <= 0) goto done;
    long end = segment_end(teach, start);
    if (end - start > avail) end = start + avail;
    somhip_som_params sp = { teach->length, teach->alpha, teach->radius, teach->alpha_type,
                             use_fixed_level, use_weights_level,
                             teach->batch > 1 || teach->batch == SOMHIP_BATCH_AUTO ? teach->batch : 1,   /* -batch auto */
                             start, end - start, first };
    if (somhip_som_train(cb, fd.ds, &sp, NULL, NULL)) { fprintf(stderr, "som_training: %s\n", somhip_last_error()); goto done; }
    feed_took(&fd, end - start);
    if (teach->snapshot && end - 1 > 0 && (end - 1) % teach->snapshot->interval == 0 && end <= teach->length) {
      if (somhip_codebook_download(cb, codes->points)) goto done;
      ifverbose(2) fprintf(stderr, "Saving snapshot, %ld iterations\n", end - 1);
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed, continuing teaching\n");
    }
    start = end;
  }
  if (somhip_codebook_download(cb, codes->points)) { fprintf(stderr, "som_training: %s\n", somhip_last_error()); goto done; }
  ret = codes;
done:
  if (cb) somhip_codebook_destroy(cb);
  feed_close(&fd);
  return ret;
}

/* ------------------------------------------------------------------ som_training on G GPUs (vsom -gpus G)
 * One process per GPU: the parent -- which has parsed the arguments and read the files, and has not
 * touched a GPU -- forks G ranks; rank r takes device r % (visible GPUs), holds every G-th 8x8-unit patch of the map
 * (somhip_codebook_create_interleaved; contiguous row blocks when a map side is not a multiple of 8) and the whole
 * data set.  Per mini-batch: somhip_batch_winner_keys on its shard (or somhip_shard_winner_begin / refine / finish with
 * the pre-filter's bounds MIN-reduced between them) -> somhip_comm_allreduce_min_keys (RCCL
 * ncclAllReduce(ncclUint64, ncclMin) on the engine's stream; the ranks get the communicator id from rank 0 over a
 * socketpair the parent made) -> somhip_som_batch_update of its own rows.  The codebook comes together only at the end
 * (X3), on rank 0, which returns it for saving.  When the ranks outnumber the GPUs (a rehearsal on one GPU: RCCL
 * refuses duplicate devices) or SOMHIP_COMM=sockets, the keys travel over the same sockets instead. */
#include <errno.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

/* whole transfers over the ranks' sockets: interrupted calls are repeated; a peer that went away is an error return
 * (MSG_NOSIGNAL: no SIGPIPE), so that the caller's own clean-up and message are what happens */
static int sock_write(int fd, const void *b, size_t n)
{
  const char *p = b;
  while (n) {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0 && errno == ENOTSOCK) k = write(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return 1;
    p += k; n -= (size_t)k;
  }
  return 0;
}
static int sock_read(int fd, void *b, size_t n)
{
  char *p = b;
  while (n) {
    ssize_t k = read(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return 1;
    p += k; n -= (size_t)k;
  }
  return 0;
}

/* what one rank does; fds: rank 0 has world-1 descriptors (peer r at [r-1]), every other rank one (to rank 0) */
static int som_training_rank(struct teach_params *teach, int rank, int world, int *fds)
{
  struct entries *codes = teach->codes, *data = teach->data;
  const long n = codes->num_entries, dim = codes->dimension, L = teach->length;
  const int auto_b = teach->batch == SOMHIP_BATCH_AUTO;     /* -batch auto: the engine's own batch boundaries */
  const long B = auto_b ? 32768 : teach->batch > 1 ? teach->batch : 4096;
  int ndev = 0, rc = 1;
  somhip_comm *comm = NULL;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  void *dkeys = NULL, *dbound = NULL, *dflag = NULL;
  int64_t *units = NULL, n_local = 0;
  float *mine = NULL;
  if (pak_rank_device(rank) < 0 || somhip_device_count(&ndev)) return 1;
  somhip_engine *en = engine();
  if (!en) return 1;
  const char *force = getenv("SOMHIP_COMM");
  const int use_rccl = force ? strcmp(force, "rccl") == 0 : ndev >= world;
  if (use_rccl) {
    char id[128];
    if (rank == 0) {
      if (somhip_comm_unique_id(id)) goto hip_fail;
      for (int r = 1; r < world; r++) if (sock_write(fds[r - 1], id, sizeof id)) goto done;
    } else if (sock_read(fds[0], id, sizeof id)) goto done;
    if (somhip_comm_create(en, id, rank, world, &comm)) goto hip_fail;
  } else if (somhip_comm_create_sockets(en, rank, world, fds, &comm)) goto hip_fail;
  ifverbose(2) fprintf(stderr, "rank %d/%d on GPU %d, keys by %s\n", rank, world, g_device, use_rccl ? "RCCL" : "host sockets");

  /* this rank's units and rows */
  const int interleaved = codes->xdim % 8 == 0 && codes->ydim % 8 == 0;
  if (interleaved) {
    if (somhip_shard_units(codes->xdim, codes->ydim, rank, world, NULL, &n_local)) goto hip_fail;
    units = malloc(sizeof(int64_t) * (n_local + 1));
    if (somhip_shard_units(codes->xdim, codes->ydim, rank, world, units, &n_local)) goto hip_fail;
  } else {
    const long per = (n + world - 1) / world, r0 = rank * per < n ? rank * per : n, r1 = r0 + per < n ? r0 + per : n;
    n_local = r1 - r0;
    units = malloc(sizeof(int64_t) * (n_local + 1));
    for (long j = 0; j < n_local; j++) units[j] = r0 + j;
  }
  if (n_local <= 0) { fprintf(stderr, "som_training: more ranks (%d) than the map can be cut into\n", world); goto done; }
  mine = malloc(sizeof(float) * n_local * dim);
  for (long j = 0; j < n_local; j++) memcpy(mine + j * dim, codes->points + units[j] * dim, sizeof(float) * dim);
  if (interleaved ? somhip_codebook_create_interleaved(en, mine, n_local, (int)dim, codes->topol, codes->neigh, codes->xdim, codes->ydim, rank, world, &cb)
                  : somhip_codebook_create(en, mine, NULL, n_local, (int)dim, codes->topol, codes->neigh, codes->xdim, codes->ydim, units[0], n, &cb)) goto hip_fail;
  if (!(ds = mirror_data(data, 0))) goto done;
  if (somhip_device_alloc(en, 8 * B, &dkeys) || somhip_device_alloc(en, 4 * B, &dbound) || somhip_device_alloc(en, 16, &dflag)) goto hip_fail;
  long exch_c = -1;
  int exch_ok = 0;

  somhip_som_params sp = { L, teach->alpha, teach->radius, teach->alpha_type, use_fixed_level, use_weights_level, B, 0, 0, 0 };
  for (long it0 = 0; it0 < L;) {                       /* batches aligned to the schedule, as somhip_som_train cuts them */
    long c = B - it0 % B < L - it0 ? B - it0 % B : L - it0;
    const long first = it0 % data->num_entries;
    if (auto_b) {
      int64_t bs, bl;
      if (somhip_som_auto_batch(&sp, n, codes->topol, codes->neigh, it0, &bs, &bl)) goto hip_fail;
      c = (long)(bs + bl - it0);
    }
    if (c != exch_c) {                                 /* every rank has to take the same path: the answers are summed once per batch length */
      /* (the exchange pays from 8 ranks on -- tools/shard_rehearsal.py; SOMHIP_SHARD_EXCHANGE=1 asks for it with fewer) */
      uint32_t f = (world >= 8 || (world > 1 && getenv("SOMHIP_SHARD_EXCHANGE"))) && somhip_shard_exchange_available(cb, ds, c) ? 1u : 0u;
      if (somhip_copy_to_device(en, dflag, &f, sizeof f) || somhip_comm_allreduce_sum_u32(comm, dflag, 1) ||
          somhip_copy_to_host(en, &f, dflag, sizeof f)) goto hip_fail;
      exch_ok = f == (uint32_t)world;
      if (exch_ok && exch_c < 0 && rank == 0) ifverbose(2) fprintf(stderr, "winner search: pre-filter bounds exchanged between the %d ranks\n", world);
      exch_c = c;
    }
    /* the winner search of the whole map: with the pre-filter's bounds going round between its levels every rank
     * re-ranks only what one GPU holding the whole map would (somhip.h, somhip_shard_winner_*) */
    if (exch_ok ? (somhip_shard_winner_begin(cb, ds, first, c, dkeys, dbound) || somhip_comm_allreduce_min_f32(comm, dbound, c) ||
                   somhip_shard_winner_refine(cb, ds, first, c, dbound) || somhip_comm_allreduce_min_f32(comm, dbound, c) ||
                   somhip_shard_winner_finish(cb, ds, first, c, dbound, dkeys))
                : somhip_batch_winner_keys(cb, ds, first, c, dkeys)) goto hip_fail;
    if (somhip_comm_allreduce_min_keys(comm, dkeys, c) || somhip_som_batch_update(cb, ds, &sp, it0, c, first, dkeys)) goto hip_fail;
    it0 += c;
  }
  if (somhip_codebook_download(cb, mine)) goto hip_fail;
  /* X3: every rank's rows, with their unit indices, to rank 0 */
  if (rank == 0) {
    for (long j = 0; j < n_local; j++) memcpy(codes->points + units[j] * dim, mine + j * dim, sizeof(float) * dim);
    for (int r = 1; r < world; r++) {
      int64_t cnt;
      if (sock_read(fds[r - 1], &cnt, sizeof cnt)) goto done;
      int64_t *u = malloc(sizeof(int64_t) * (cnt + 1));
      float *rows = malloc(sizeof(float) * (cnt + 1) * dim);
      const int bad = sock_read(fds[r - 1], u, sizeof(int64_t) * cnt) || sock_read(fds[r - 1], rows, sizeof(float) * cnt * dim);
      for (long j = 0; !bad && j < cnt; j++) memcpy(codes->points + u[j] * dim, rows + j * dim, sizeof(float) * dim);
      free(u); free(rows);
      if (bad) goto done;
    }
  } else if (sock_write(fds[0], &n_local, sizeof n_local) || sock_write(fds[0], units, sizeof(int64_t) * n_local) ||
             sock_write(fds[0], mine, sizeof(float) * n_local * dim)) goto done;
  rc = 0;
  goto done;
hip_fail:
  fprintf(stderr, "som_training (rank %d): %s\n", rank, somhip_last_error());
done:
  if (dkeys) somhip_device_free(en, dkeys);
  if (dbound) somhip_device_free(en, dbound);
  if (dflag) somhip_device_free(en, dflag);
  if (ds) somhip_dataset_destroy(ds);
  if (cb) somhip_codebook_destroy(cb);
  if (comm) somhip_comm_destroy(comm);
  free(units); free(mine);
  return rc;
}

/* Ends the ranks still alive: SIGTERM, a grace period, SIGKILL, and reaps every one of them (a rank that sits in a
 * collective whose peer is gone -- ncclAllReduce / hipStreamSynchronize -- never returns by itself). */
static void ranks_kill_rest(pid_t *pid, int world)
{
  int alive = 0;
  for (int r = 0; r < world; r++) if (pid[r] > 0) { kill(pid[r], SIGTERM); alive++; }
  for (int tick = 0; alive && tick < 50; tick++) {           /* up to 5 s */
    for (int r = 0; r < world; r++)
      if (pid[r] > 0 && waitpid(pid[r], NULL, WNOHANG) == pid[r]) { pid[r] = 0; alive--; }
    if (alive) { struct timespec ts = {0, 100000000}; nanosleep(&ts, NULL); }
  }
  for (int r = 0; r < world; r++)
    if (pid[r] > 0) { kill(pid[r], SIGKILL); waitpid(pid[r], NULL, 0); pid[r] = 0; }
}

/* One process per GPU: forks `world` ranks of the calling process -- which has read its files and has NOT touched a GPU
 * yet -- and runs rank_main(rank, world, fds, arg) in each; fds: rank 0 gets world-1 socket descriptors (peer r at
 * [r-1]), every other rank one (to rank 0).  Rank r uses device r % (visible GPUs) (pak_rank_device).  Returns 0 when
 * every rank returned 0.  The ranks are reaped in the order in which they end; the first one that fails (non-zero exit
 * or a signal) -- or a fork() that fails half way -- ends the others (ranks_kill_rest) and the call returns 1: no rank
 * is left behind in a collective, no orphan keeps a GPU.  Children are only ever started fresh or killed, never
 * re-executed. */
int pak_run_ranks(int world, int (*rank_main)(int rank, int world, int *fds, void *arg), void *arg)
{
  if (g_engine) { fprintf(stderr, "the ranks must be started before this process uses a GPU\n"); return 1; }
  if (world < 1 || world > 64) { fprintf(stderr, "-gpus %d?\n", world); return 1; }
  int (*sv)[2] = malloc(sizeof(int[2]) * (world > 1 ? world - 1 : 1));
  for (int r = 1; r < world; r++)
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv[r - 1])) {
      perror("socketpair");
      for (int q = 1; q < r; q++) { close(sv[q - 1][0]); close(sv[q - 1][1]); }
      free(sv);
      return 1;
    }
  pid_t *pid = calloc((size_t)world, sizeof(pid_t));
  fflush(NULL);
  int bad = 0;
  for (int r = 0; r < world && !bad; r++) {
    pid[r] = fork();
    if (pid[r] < 0) { perror("fork"); pid[r] = 0; bad = 1; break; }
    if (pid[r] == 0) {
      signal(SIGPIPE, SIG_IGN);      /* a peer that went away is an error return of the write (host_comm.inc), not a signal */
      int *fds = malloc(sizeof(int) * (world > 1 ? world - 1 : 1));
      for (int q = 1; q < world; q++) {
        if (r == 0) { fds[q - 1] = sv[q - 1][0]; close(sv[q - 1][1]); }
        else if (q == r) { fds[0] = sv[q - 1][1]; close(sv[q - 1][0]); }
        else { close(sv[q - 1][0]); close(sv[q - 1][1]); }
      }
      const int rc = rank_main(r, world, fds, arg);
      pak_shutdown();
      fflush(NULL);
      _exit(rc ? 1 : 0);
    }
  }
  for (int r = 1; r < world; r++) { close(sv[r - 1][0]); close(sv[r - 1][1]); }
  int left = 0;
  for (int r = 0; r < world; r++) if (pid[r] > 0) left++;
  while (!bad && left > 0) {
    int st = 0;
    const pid_t w = waitpid(-1, &st, 0);
    if (w < 0) { if (errno == EINTR) continue; bad = 1; break; }
    int r = 0;
    while (r < world && pid[r] != w) r++;
    if (r == world) continue;                                  /* some other child of the host program */
    pid[r] = 0; left--;
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      if (WIFSIGNALED(st)) fprintf(stderr, "rank %d ended by signal %d; stopping the other ranks\n", r, WTERMSIG(st));
      else fprintf(stderr, "rank %d failed; stopping the other ranks\n", r);
      bad = 1;
    }
  }
  if (bad) ranks_kill_rest(pid, world);
  free(sv); free(pid);
  return bad;
}
int pak_rank_device(int rank)
{
  int ndev = 0;
  if (somhip_device_count(&ndev) || ndev < 1) { fprintf(stderr, "%s\n", somhip_last_error()); return -1; }
  g_device = rank % ndev;
  return g_device;
}
int pak_sock_write(int fd, const void *b, size_t n) { return sock_write(fd, b, n); }
int pak_sock_read(int fd, void *b, size_t n) { return sock_read(fd, b, n); }

struct som_multi { struct teach_params *teach; int (*after)(struct teach_params *, void *); void *arg; };
static int som_multi_rank(int rank, int world, int *fds, void *p)
{
  struct som_multi *m = p;
  int rc = som_training_rank(m->teach, rank, world, fds);
  if (rc == 0 && rank == 0 && m->after) rc = m->after(m->teach, m->arg);
  return rc;
}
/* vsom -gpus G: the ranks train the sharded map; rank 0 also runs after(teach, arg) -- the tool's "save the codebook" */
int som_training_multi(struct teach_params *teach, int gpus, int (*after)(struct teach_params *, void *), void *arg)
{
  if (set_som_params(teach)) { fprintf(stderr, "som_training: can't set SOM parameters\n"); return 1; }
  if (!teach->data || teach->data->num_entries <= 0) { fprintf(stderr, "som_training: can't get data\n"); return 1; }
  if (teach->data->dimension != teach->codes->dimension) {
    fprintf(stderr, "code dimension (%d) != data dimension (%d)\n", teach->codes->dimension, teach->data->dimension);
    return 1;
  }
  if (teach->data->random_order && teach->data->buffer > 0 && teach->data->buffer < teach->data->num_entries) {
    fprintf(stderr, "som_training: -buffer with -rand is not available with -gpus\n");
    return 1;
  }
  if (teach->snapshot) fprintf(stderr, "som_training: snapshots are not written with -gpus\n");
  struct som_multi m = { teach, after, arg };
  return pak_run_ranks(gpus, som_multi_rank, &m);
}

static struct entries *lvq_training(struct teach_params *teach, int kind, float winlen, float epsilon,
                                    float *talpha, const char *who)
{
  struct entries *codes = teach->codes, *data = teach->data;
  if (!data || data->num_entries <= 0) { fprintf(stderr, "%s: can't get data\n", who); return NULL; }
  somhip_codebook *cb = mirror_codes(codes, 1);
  struct feed fd;
  struct entries *ret = NULL;
  if (feed_open(&fd, data, 1) || !cb) goto done;
  for (long start = 0; start < teach->length;) {
    long first, avail = feed_avail(&fd, start, &first);
    if (avail <= 0) goto done;
    long end = segment_end(teach, start);
    if (end - start > avail) end = start + avail;
    somhip_lvq_params lp = { kind, teach->length, teach->alpha, teach->alpha_type, winlen, epsilon,
                             start, end - start, first };
    if (somhip_lvq_train(cb, fd.ds, &lp, talpha, NULL, NULL)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); goto done; }
    feed_took(&fd, end - start);
    if (teach->snapshot && end - 1 > 0 && (end - 1) % teach->snapshot->interval == 0 && end <= teach->length) {
      if (somhip_codebook_download(cb, codes->points)) goto done;
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed\n");
    }
    start = end;
  }
  if (somhip_codebook_download(cb, codes->points)) goto done;
  ret = codes;
done:
  if (cb) somhip_codebook_destroy(cb);
  feed_close(&fd);
  return ret;
}
struct entries *lvq1_training(struct teach_params *t) { return lvq_training(t, SOMHIP_LVQ1, 0, 0, NULL, "lvq1_training"); }
struct entries *lvq2_training(struct teach_params *t, float winlen) { return lvq_training(t, SOMHIP_LVQ2, winlen, 0, NULL, "lvq2_training"); }
struct entries *lvq3_training(struct teach_params *t, float eps, float winlen) { return lvq_training(t, SOMHIP_LVQ3, winlen, eps, NULL, "lvq3_training"); }

struct entries *olvq1_training(struct teach_params *teach, const char *infile, const char *outfile)  /* lvq_rout.c:584-697 */
{
  long noc = teach->codes->num_entries;
  float *talpha = malloc(sizeof(float) * (noc + 1));
  float alpha = teach->alpha;
  if (alpha == 0.0f) {                                        /* :615-622 */
    if (!alpha_read(talpha, noc, infile)) {
      alpha = 0.3f;
      for (long i = 0; i < noc; i++) talpha[i] = alpha;
    }
  } else {
    for (long i = 0; i < noc; i++) talpha[i] = alpha;
  }
  float keep = teach->alpha;
  teach->alpha = alpha;                                       /* the clamp of :671 is the local `alpha` */
  struct entries *r = lvq_training(teach, SOMHIP_OLVQ1, 0, 0, talpha, "olvq1_training");
  teach->alpha = keep;
  if (r) alpha_write(talpha, noc, outfile);                  /* :694 */
  free(talpha);
  return r;
}

/* ------------------------------------------------------------------ lvq*_training on G GPUs (lvqtrain -gpus G)
 * The codebook is cut into contiguous row blocks, one per rank; every rank holds the data.  Per batch of <= 1024
 * iterations (include/somhip.h, "lvq*_training over a ROW-SHARDED codebook"): each rank's 8 nearest rows per sample ->
 * all-gather + merge -> labels / rates / rows of the listed candidates the rank owns -> all-reduce(SUM) as integers ->
 * every rank walks the batch (same decisions everywhere) and commits the rows it owns.  Exactly the online result.
 * Collectives: RCCL when every rank has its own GPU, the parent's socketpairs otherwise (somhip_comm). */
struct lvq_multi {
  struct teach_params *teach; int kind; float winlen, epsilon, clamp; float *talpha;
  int (*after)(struct teach_params *, void *); void *arg;
};

static int lvq_training_rank(int rank, int world, int *fds, void *pp)
{
  struct lvq_multi *m = pp;
  struct teach_params *teach = m->teach;
  struct entries *codes = teach->codes, *data = teach->data;
  const long n = codes->num_entries, dim = codes->dimension, L = teach->length;
  const long per = (n + world - 1) / world, r0 = rank * per < n ? rank * per : n, r1 = r0 + per < n ? r0 + per : n, nl = r1 - r0;
  const int knn = m->kind >= SOMHIP_LVQ2 ? 2 : 1, XR = 4, BMAX = 1024;
  const long d4 = (dim + 3) / 4;
  int ndev = 0, rc = 1;
  somhip_comm *comm = NULL;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  void *dloc = NULL, *dall = NULL, *dkeys = NULL, *dlab = NULL, *dta = NULL, *drows = NULL;
  float *mine = NULL;
  if (nl <= 0) { fprintf(stderr, "lvq training: more ranks (%d) than code vectors\n", world); return 1; }
  if (pak_rank_device(rank) < 0 || somhip_device_count(&ndev)) return 1;
  somhip_engine *en = engine();
  if (!en) return 1;
  const char *force = getenv("SOMHIP_COMM");
  const int use_rccl = force ? strcmp(force, "rccl") == 0 : ndev >= world;
  if (use_rccl) {
    char id[128];
    if (rank == 0) {
      if (somhip_comm_unique_id(id)) goto hip_fail;
      for (int r = 1; r < world; r++) if (sock_write(fds[r - 1], id, sizeof id)) goto done;
    } else if (sock_read(fds[0], id, sizeof id)) goto done;
    if (somhip_comm_create(en, id, rank, world, &comm)) goto hip_fail;
  } else if (somhip_comm_create_sockets(en, rank, world, fds, &comm)) goto hip_fail;

  int32_t *lab = first_labels(codes);
  if (somhip_codebook_create(en, codes->points + r0 * dim, lab + r0, nl, (int)dim, TOPOL_LVQ, 0, 0, 0, r0, n, &cb)) { free(lab); goto hip_fail; }
  free(lab);
  if (m->kind == SOMHIP_OLVQ1 && somhip_lvq_rates_upload(cb, m->talpha + r0)) goto hip_fail;
  if (!(ds = mirror_data(data, 1))) goto done;
  if (somhip_device_alloc(en, 8 * 8 * BMAX, &dloc) || somhip_device_alloc(en, (int64_t)8 * 8 * BMAX * world, &dall) ||
      somhip_device_alloc(en, 8 * 8 * BMAX, &dkeys) || somhip_device_alloc(en, 4 * 8 * BMAX, &dlab) ||
      somhip_device_alloc(en, 4 * 8 * BMAX, &dta) || somhip_device_alloc(en, (int64_t)BMAX * XR * d4 * 16, &drows)) goto hip_fail;

  somhip_lvq_params lp = { m->kind, L, m->clamp, teach->alpha_type, m->winlen, m->epsilon, 0, 0, 0 };
  long B = 256;
  for (long it0 = 0; it0 < L;) {
    const long c = B < L - it0 ? B : L - it0, first = it0 % data->num_entries;
    int64_t done = 0;
    if (somhip_batch_topk_keys(cb, ds, first, c, 8, knn == 2 ? SOMHIP_TIE_KNN : SOMHIP_TIE_FIRST, dloc) ||
        somhip_comm_allgather(comm, dloc, dall, 8 * 8 * c) ||
        somhip_merge_topk_keys(en, dall, world, c, 8, dkeys) ||
        somhip_lvq_batch_candidates(cb, c, m->kind, dkeys, XR, dlab, m->kind == SOMHIP_OLVQ1 ? dta : NULL, drows) ||
        somhip_comm_allreduce_sum_u32(comm, dlab, 8 * c) ||
        (m->kind == SOMHIP_OLVQ1 && somhip_comm_allreduce_sum_u32(comm, dta, 8 * c)) ||
        somhip_comm_allreduce_sum_u32(comm, drows, c * XR * d4 * 4) ||
        somhip_lvq_batch_apply(cb, ds, &lp, it0, c, first, dkeys, dlab, m->kind == SOMHIP_OLVQ1 ? dta : NULL, drows, XR, &done, NULL, NULL))
      goto hip_fail;
    if (done <= 0 || done > c) { fprintf(stderr, "lvq training: batch made no progress\n"); goto done; }
    it0 += done;
    B = done == c ? (2 * B < BMAX ? 2 * B : BMAX) : (done + done / 4 + 8 > 32 ? (done + done / 4 + 8 < BMAX ? done + done / 4 + 8 : BMAX) : 32);
  }
  /* X3: every rank's rows (and OLVQ1 rates) to rank 0 */
  mine = malloc(sizeof(float) * nl * (dim + 1));
  if (somhip_codebook_download(cb, mine)) goto hip_fail;
  if (m->kind == SOMHIP_OLVQ1 && somhip_lvq_rates_download(cb, mine + nl * dim)) goto hip_fail;
  if (rank == 0) {
    memcpy(codes->points, mine, sizeof(float) * nl * dim);
    if (m->kind == SOMHIP_OLVQ1) memcpy(m->talpha, mine + nl * dim, sizeof(float) * nl);
    for (int r = 1; r < world; r++) {
      const long q0 = r * per < n ? r * per : n, q1 = q0 + per < n ? q0 + per : n, nq = q1 - q0;
      float *buf = malloc(sizeof(float) * (nq + 1) * (dim + 1));
      const int bad = sock_read(fds[r - 1], buf, sizeof(float) * nq * (dim + 1));
      if (!bad) {
        memcpy(codes->points + q0 * dim, buf, sizeof(float) * nq * dim);
        if (m->kind == SOMHIP_OLVQ1) memcpy(m->talpha + q0, buf + nq * dim, sizeof(float) * nq);
      }
      free(buf);
      if (bad) goto done;
    }
    rc = m->after ? m->after(teach, m->arg) : 0;
  } else {
    rc = sock_write(fds[0], mine, sizeof(float) * nl * (dim + 1));
  }
  goto done;
hip_fail:
  fprintf(stderr, "lvq training (rank %d): %s\n", rank, somhip_last_error());
done:
  { void *bufs[6] = { dloc, dall, dkeys, dlab, dta, drows }; for (int k = 0; k < 6; k++) if (bufs[k]) somhip_device_free(en, bufs[k]); }
  if (ds) somhip_dataset_destroy(ds);
  if (cb) somhip_codebook_destroy(cb);
  if (comm) somhip_comm_destroy(comm);
  free(mine);
  return rc;
}

/* lvqtrain -gpus G.  talpha: OLVQ1's rates, [noc], filled in by the caller as lvq_rout.c:614-627 does and updated in
 * rank 0's copy before after(teach, arg) runs there (save the codebook, write the .lra file). */
int lvq_training_multi(struct teach_params *teach, int kind, float winlen, float epsilon, float clamp, float *talpha, int gpus,
                       int (*after)(struct teach_params *, void *), void *arg)
{
  if (!teach->data || teach->data->num_entries <= 0) { fprintf(stderr, "lvq training: can't get data\n"); return 1; }
  if (teach->data->random_order && teach->data->buffer > 0 && teach->data->buffer < teach->data->num_entries) {
    fprintf(stderr, "lvq training: -buffer with -rand is not available with -gpus\n");
    return 1;
  }
  if (teach->data->masks) { fprintf(stderr, "lvq training: masked samples are not supported by the LVQ loops of the engine\n"); return 1; }
  if (teach->snapshot) fprintf(stderr, "lvq training: snapshots are not written with -gpus\n");
  struct lvq_multi m = { teach, kind, winlen, epsilon, clamp, talpha, after, arg };
  return pak_run_ranks(gpus, lvq_training_rank, &m);
}

int find_all_winners(struct teach_params *teach, int32_t *index, float *diff, int32_t *ret)
{
  somhip_codebook *cb = mirror_codes(teach->codes, 0);
  somhip_dataset *ds = mirror_data(teach->data, 0);
  int rc = 1;
  if (cb && ds) {
    rc = somhip_find_winners(cb, ds, 0, teach->data->num_entries, 1, SOMHIP_TIE_FIRST, index, diff, ret);
    if (rc) fprintf(stderr, "%s\n", somhip_last_error());
  }
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  return rc;
}

/* randinit_codes (som_rout.c:34-162): every component uniform in the bounding box of the data
 * (unmasked components only), orand() drawn unit by unit, component by component. */
struct entries *randinit_codes(struct entries *data, int topol, int neigh, int xdim, int ydim)
{
  int dim = data->dimension;
  long noc = (long)xdim * ydim;
  /* the reference seeds its maximum with FLT_MIN (the smallest positive float), som_rout.c:108-111 */
  float *hi = malloc(sizeof(float) * dim), *lo = malloc(sizeof(float) * dim);
  long *cnt = calloc(dim, sizeof(long));
  for (int i = 0; i < dim; i++) { hi[i] = FLT_MIN; lo[i] = FLT_MAX; }
  if (data->is_virtual) {                              /* the bounding box of a generated source is one pass in HBM */
    somhip_dataset *ds = mirror_data(data, 0);
    int64_t *c64 = malloc(sizeof(int64_t) * dim);
    float *dlo = malloc(sizeof(float) * dim), *dhi = malloc(sizeof(float) * dim);
    if (!ds || somhip_column_minmax(ds, dlo, dhi, c64)) {
      fprintf(stderr, "randinit_codes: %s\n", somhip_last_error());
      free(hi); free(lo); free(cnt); free(c64); free(dlo); free(dhi);
      if (ds) somhip_dataset_destroy(ds);
      return NULL;
    }
    for (int i = 0; i < dim; i++) {
      cnt[i] = (long)c64[i];
      if (cnt[i] > 0) { if (hi[i] < dhi[i]) hi[i] = dhi[i]; if (lo[i] > dlo[i]) lo[i] = dlo[i]; }
    }
    somhip_dataset_destroy(ds);
    free(c64); free(dlo); free(dhi);
  }
  for (long r = 0; !data->is_virtual && r < data->num_entries; r++) {
    struct data_entry *e = &data->rows[r];
    for (int i = 0; i < dim; i++)
      if (!(e->mask && e->mask[i])) {
        cnt[i]++;
        if (hi[i] < e->points[i]) hi[i] = e->points[i];
        if (lo[i] > e->points[i]) lo[i] = e->points[i];
      }
  }
  for (int i = 0; i < dim; i++)
    if (cnt[i] == 0) fprintf(stderr, "randinit_codes: warning! component %d has no data, using 0.0\n", i + 1);
  struct entries *codes = calloc(1, sizeof *codes);
  codes->dimension = (short)dim; codes->topol = (short)topol; codes->neigh = (short)neigh;
  codes->xdim = (short)xdim; codes->ydim = (short)ydim; codes->num_entries = noc;
  codes->points = malloc(sizeof(float) * noc * dim);
  codes->rows = calloc(noc, sizeof(struct data_entry));
  for (long k = 0; k < noc; k++) {
    codes->rows[k].points = codes->points + k * dim;
    for (int i = 0; i < dim; i++)                  /* som_rout.c:140-150 */
      codes->rows[k].points[i] = cnt[i] > 0 ? lo[i] + (hi[i] - lo[i]) * ((float)orand() / 32768.0) : 0.0;
  }
  free(hi); free(lo); free(cnt);
  return codes;
}

/* lininit_codes (som_rout.c:322-429) with find_eigenvectors (:211-320): the map is laid out on the
 * plane spanned by the two principal axes of the data.  The two passes over the data (mean, upper
 * triangle of the centred product sums -- O(n dim^2)) run on the MI355X engine with the reference's
 * fp32 accumulation order; the 10-step two-vector power iteration on the dim x dim matrix is host
 * work, written with the reference's float / double mix so that every rounding falls where it does
 * there. */
static void normalize_f(float *v, int n)              /* som_rout.c:166-174 */
{
  float sum = 0.0;
  for (int j = 0; j < n; j++) sum += v[j] * v[j];
  sum = sqrt(sum);
  for (int j = 0; j < n; j++) v[j] /= sum;
}
static float dotprod_f(const float *v, const float *w, int n)     /* :177-184 */
{
  float sum = 0.0;
  for (int j = 0; j < n; j++) sum += v[j] * w[j];
  return sum;
}
static void gram_schmidt_f(float *v, int n, int e)    /* :187-209 */
{
  float *w = malloc(sizeof(float) * n * e);
  for (int i = 0; i < e; i++) {
    for (int t = 0; t < n; t++) {
      float sum = v[i * n + t];
      for (int j = 0; j < i; j++)
        for (int p = 0; p < n; p++) sum -= w[j * n + t] * w[j * n + p] * v[i * n + p];
      w[i * n + t] = sum;
    }
    normalize_f(w + i * n, n);
  }
  memcpy(v, w, sizeof(float) * n * e);
  free(w);
}

struct entries *lininit_codes(struct entries *data, int topol, int neigh, int xdim, int ydim)
{
  int n = data->dimension;
  long k = data->num_entries, noc = (long)xdim * ydim;
  float *m = malloc(sizeof(float) * n), *r = malloc(sizeof(float) * n * n);
  float *u = malloc(sizeof(float) * 2 * n), *v = malloc(sizeof(float) * 2 * n);
  int64_t *k2 = malloc(sizeof(int64_t) * n);
  float mu[2];
  struct entries *codes = NULL;
  somhip_dataset *ds = mirror_data(data, 0);
  if (!ds) goto fail;
  if (somhip_column_sums(ds, m, k2)) { fprintf(stderr, "%s\n", somhip_last_error()); goto fail; }
  if (k < 3) goto fail;                                 /* :256 */
  for (int i = 0; i < n; i++) m[i] /= k2[i];
  if (somhip_centered_products(ds, m, r)) { fprintf(stderr, "%s\n", somhip_last_error()); goto fail; }
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) r[j * n + i] = r[i * n + j] /= k;
  for (int i = 0; i < 2; i++) {
    for (int j = 0; j < n; j++) u[i * n + j] = orand() / 16384.0 - 1.0;
    normalize_f(u + i * n, n);
    mu[i] = 1.0;
  }
  for (int it = 0; it < 10; it++) {
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < n; j++) v[i * n + j] = mu[i] * dotprod_f(r + j * n, u + i * n, n) + u[i * n + j];
    gram_schmidt_f(v, n, 2);
    float sum = 0.0;                                    /* not reset between the two vectors (:300-306) */
    for (int i = 0; i < 2; i++) {
      for (int j = 0; j < n; j++) sum += fabs(v[i * n + j] / dotprod_f(r + j * n, v + i * n, n));
      mu[i] = sum / n;
    }
    memcpy(u, v, sizeof(float) * 2 * n);
  }
  if (mu[0] == 0.0 || mu[1] == 0.0) goto fail;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < n; j++) u[i * n + j] /= sqrt(mu[i]);

  codes = calloc(1, sizeof *codes);
  codes->dimension = (short)n; codes->topol = (short)topol; codes->neigh = (short)neigh;
  codes->xdim = (short)xdim; codes->ydim = (short)ydim; codes->num_entries = noc;
  codes->points = malloc(sizeof(float) * noc * n);
  codes->rows = calloc(noc, sizeof(struct data_entry));
  for (long index = 0; index < noc; index++) {          /* :405-421 */
    float xf = 4.0 * (float)(index % xdim) / (xdim - 1.0) - 2.0;
    float yf = 4.0 * (float)(index / xdim) / (ydim - 1.0) - 2.0;
    float *pt = codes->rows[index].points = codes->points + index * n;
    for (int i = 0; i < n; i++) pt[i] = m[i] + xf * u[i] + yf * u[n + i];
  }
fail:
  if (!codes) fprintf(stderr, "lininit_codes: Can't find eigenvectors\n");
  if (ds) somhip_dataset_destroy(ds);
  free(m); free(r); free(u); free(v); free(k2);
  return codes;
}

/* k nearest codes of every data row (find_winner_knn, lvq_pak.c:152-221; knn = 1 is
 * find_winner_euc): index/diff [n][knn], nearest first, ties in the reference's order. */
int find_all_knn(struct entries *codes, struct entries *data, int knn, int32_t *index, float *diff)
{
  if (knn < 1) knn = 1;
  if (knn > 8) { fprintf(stderr, "this engine finds at most 8 nearest neighbours (-knn %d)\n", knn); return 1; }
  somhip_codebook *cb = mirror_codes(codes, 0);
  somhip_dataset *ds = mirror_data(data, 0);
  int rc = 1;
  if (cb && ds) {
    rc = somhip_find_winners(cb, ds, 0, data->num_entries, knn, knn >= 2 ? SOMHIP_TIE_KNN : SOMHIP_TIE_FIRST,
                             index, diff, NULL);
    if (rc) fprintf(stderr, "%s\n", somhip_last_error());
  }
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  return rc;
}

/* correct_by_knn (lvq_rout.c:38-78) for every row of `data` against `data` itself: the majority
 * label (head of the hit list built nearest-first) equals the row's own first label. */
unsigned char *knn_correct_all(struct entries *data, int knn)
{
  long n = data->num_entries;
  if (knn < 1) knn = 1;
  int32_t *idx = malloc(sizeof(int32_t) * n * knn);
  float *diff = malloc(sizeof(float) * n * knn);
  unsigned char *ok = calloc(n, 1);
  if (find_all_knn(data, data, knn, idx, diff)) { free(idx); free(diff); free(ok); return NULL; }
  for (long r = 0; r < n; r++) {
    struct hitlist *hits = new_hitlist();
    int found = 1;
    for (int k = 0; k < knn; k++) {
      long w = idx[r * knn + k];
      if (w < 0) { found = 0; break; }
      add_hit(hits, get_entry_label(&data->rows[w]));
    }
    if (!found) { fprintf(stderr, "correct_by_knn: can't find winners\n"); ok[r] = 1; }   /* -1 is "true" at :182 */
    else ok[r] = hits->entries > 0 && hits->label[0] == get_entry_label(&data->rows[r]);
    free_hitlist(hits);
  }
  free(idx); free(diff);
  return ok;
}

/* a new entries block holding copies of the given rows of src (copy_entries + copy_entry,
 * datafile.c): header fields of src, vectors, masks and all labels */
struct entries *pick_rows(struct entries *src, const long *rows, long n)
{
  struct entries *e = calloc(1, sizeof *e);
  int dim = src->dimension;
  e->dimension = src->dimension; e->topol = src->topol; e->neigh = src->neigh;
  e->xdim = src->xdim; e->ydim = src->ydim; e->num_entries = n;
  e->points = malloc(sizeof(float) * (n > 0 ? n : 1) * dim);
  e->rows = calloc(n > 0 ? n : 1, sizeof(struct data_entry));
  if (src->masks) e->masks = calloc((n > 0 ? n : 1) * dim, 1);
  for (long k = 0; k < n; k++) {
    struct data_entry *s = &src->rows[rows[k]], *d = &e->rows[k];
    d->points = e->points + k * dim;
    memcpy(d->points, s->points, sizeof(float) * dim);
    if (e->masks && s->mask) { d->mask = e->masks + k * dim; memcpy(d->mask, s->mask, dim); }
    for (int l = 0; l < s->num_labs; l++) add_entry_label(e, k, s->labels[l]);
    d->weight = s->weight;
  }
  if (src->weights) {
    e->weights = malloc(sizeof(short) * (n + 1));
    for (long k = 0; k < n; k++) e->weights[k] = src->weights[rows[k]];
  }
  if (src->fixed_xy) {
    e->fixed_xy = malloc(sizeof(short) * 2 * (n + 1));
    for (long k = 0; k < n; k++) {
      e->fixed_xy[2 * k] = src->fixed_xy[2 * rows[k]]; e->fixed_xy[2 * k + 1] = src->fixed_xy[2 * rows[k] + 1];
      e->rows[k].fixed = e->fixed_xy[2 * k] >= 0 ? (struct fixpoint *)(e->fixed_xy + 2 * k) : NULL;
    }
  }
  return e;
}

float find_qerror(struct teach_params *teach)                 /* som_rout.c:678-731 */
{
  if (set_som_params(teach)) { fprintf(stderr, "find_qerror: can't set SOM parameters\n"); return -1; }
  long n = teach->data->num_entries;
  if (n <= 0) { fprintf(stderr, "find_qerror: can't get data\n"); return -1.0f; }
  int32_t *idx = malloc(sizeof(int32_t) * n), *ret = malloc(sizeof(int32_t) * n);
  float *diff = malloc(sizeof(float) * n);
  float qerror = 0.0f;
  if (find_all_winners(teach, idx, diff, ret)) qerror = -1.0f;
  else
    for (long i = 0; i < n; i++) {
      if (ret[i] == 0) continue;                              /* ignore empty vectors, :712 */
      qerror += sqrt((double)diff[i]);                        /* float accumulator, :715 */
    }
  free(idx); free(ret); free(diff);
  return qerror;
}

float find_qerror2(struct teach_params *teach)                /* som_rout.c:823-885 */
{
  if (set_som_params(teach)) { fprintf(stderr, "find_qerror2: can't set SOM parameters\n"); return -1; }
  long n = teach->data->num_entries;
  if (n <= 0) { fprintf(stderr, "find_qerror2: can't get data\n"); return -1.0f; }
  ifverbose(3) fprintf(stderr, "qmode 1, %s neighbourhood\n", teach->codes->neigh == NEIGH_GAUSSIAN ? "gaussian" : "bubble");
  float *q = malloc(sizeof(float) * n);
  int32_t *ret = malloc(sizeof(int32_t) * n);
  somhip_codebook *cb = mirror_codes(teach->codes, 0);
  somhip_dataset *ds = mirror_data(teach->data, 0);
  float qerror = -1.0f;
  if (cb && ds) {
    if (somhip_qerror2(cb, ds, teach->radius, 0, n, q, ret)) fprintf(stderr, "%s\n", somhip_last_error());
    else {
      qerror = 0.0f;
      for (long i = 0; i < n; i++)
        if (ret[i]) qerror += q[i];                           /* ignore empty vectors, :858; float sum :864 */
    }
  }
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  free(q); free(ret);
  return qerror;
}
