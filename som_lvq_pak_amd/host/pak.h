/* pak.h -- host side of the MI355X SOM/LVQ tools, in C.
 *
 * The command-line tools here (vsom, lvqtrain = lvq1/olvq1/lvq2/lvq3, qerror, accuracy, vcal)
 * keep the flags, file formats and output text of SOM_PAK/LVQ_PAK 3.2 so existing .dat/.cod
 * files and scripts drop in; the hot path goes through the C ABI of libsomhip.so
 * (include/somhip.h).  The plug-in surface keeps the reference's names and meaning
 * (struct teach_params and its function-pointer types, lvq_pak.h:131-148,186-204; registry
 * selected with -selfuncs, datafile.c:1207-1243) so code written against that interface still
 * reads the same -- but the storage behind it is dense (one float block per file, rows are
 * views), because that is what a GPU mirror wants.
 */
#ifndef PAK_H
#define PAK_H

#include <stdio.h>
#include <stdint.h>
#include "somhip.h"

/* ids as in the reference (lvq_pak.h:209-224) */
#define TOPOL_UNKNOWN 0
#define TOPOL_DATA 1
#define TOPOL_LVQ 2
#define TOPOL_HEXA 3
#define TOPOL_RECT 4
#define NEIGH_UNKNOWN 0
#define NEIGH_BUBBLE 1
#define NEIGH_GAUSSIAN 2
#define ALPHA_UNKNOWN 0
#define ALPHA_LINEAR 1
#define ALPHA_INVERSE_T 2
#define LABEL_EMPTY 0

struct fixpoint { short xfix, yfix; };

/* one row: a view into its file's dense blocks */
struct data_entry {
  float *points;
  int *labels;          /* num_labs label ids */
  short num_labs;
  short weight;
  char *mask;           /* NULL or dim flags, nonzero = ignore component */
  struct fixpoint *fixed;
};

struct entries {
  short dimension, topol, neigh, xdim, ydim;
  long num_entries;
  struct data_entry *rows;      /* [num_entries] views */
  float *points;                /* [num_entries][dimension] */
  char *masks;                  /* NULL if no row has a masked component */
  short *fixed_xy;              /* [num_entries][2], -1 = none (NULL if none at all) */
  short *weights;               /* [num_entries] (NULL if none) */
  int labels_needed;
  long buffer;                  /* > 0 with random_order: -buffer N -rand, rows are presented buffer by buffer, */
  int random_order;             /*   each buffer reshuffled when it is (re)loaded (datafile.c:237-344)            */
  int is_virtual;               /* a `gen:` source kept as its specification: rows exist only on the device */
  unsigned long long gen_seed;  /*   (vsom / qerror / randinit / lininit, see pak_gen_virtual_ok); pak_materialize() */
  int gen_k;                    /*   makes the host rows when something needs them after all                        */
  void *userdata;               /* device mirror handle (as lvq_pak.h:112) */
  unsigned long mirror_generation;   /* host-row generation the mirror was uploaded at (paklib.c, per-sample surface) */
};

struct winner_info { long index; struct data_entry *winner; float diff; };

struct teach_params;
typedef void  NEIGH_ADAPT(struct teach_params *, struct data_entry *sample, int bx, int by, float radius, float alpha);
typedef void  VECTOR_ADAPT(struct data_entry *c, struct data_entry *s, int d, float a);
typedef float MAPDIST_FUNCTION(int bx, int by, int tx, int ty);
typedef float DIST_FUNCTION(struct data_entry *v1, struct data_entry *v2, int dim);
typedef int   WINNER_FUNCTION(struct entries *codes, struct data_entry *sample, struct winner_info *w, int knn);
typedef float ALPHA_FUNC(long iter, long length, float alpha);

struct snapshot_info { long interval; char *filename; int counter; };

struct teach_params {
  short topol, neigh, alpha_type;
  MAPDIST_FUNCTION *mapdist;
  DIST_FUNCTION *dist;
  NEIGH_ADAPT *neigh_adapt;
  VECTOR_ADAPT *vector_adapt;
  WINNER_FUNCTION *winner;
  ALPHA_FUNC *alpha_func;
  float radius, alpha;
  long length;
  int knn;
  struct entries *codes, *data;
  struct snapshot_info *snapshot;
  long batch;                   /* new: -batch B (1 = the reference's online schedule) */
};

/* ---- arguments, verbosity (same conventions as lvq_pak.c:583-660) ---- */
#define ALWAYS 1
#define OPTION 0
#define OPTION2 2
char *extract_parameter(int argc, char **argv, const char *param, int when);
long oatoi(const char *s, long def);
float oatof(const char *s, float def);
int global_options(int argc, char **argv);
extern int verbose_level;
#define ifverbose(l) if (verbose_level >= (l))
const char *pak_progname(const char *argv0);

/* ---- labels (1-based ids in order of first appearance; labels.c:75) ---- */
int find_conv_to_ind(const char *str);
const char *find_conv_to_lab(int ind);
int number_of_labels(void);
#define get_entry_label(e) ((e)->num_labs > 0 ? (e)->labels[0] : LABEL_EMPTY)

/* frequency-ordered hit lists (labels.c:370-407): new labels go last, a label moves up while
 * its predecessor has a strictly smaller count */
struct hitlist { long *label, *freq; long entries, cap; };
struct hitlist *new_hitlist(void);
void free_hitlist(struct hitlist *);
long add_hit(struct hitlist *, long label);
long hitlist_label_freq(struct hitlist *, long label);

/* ---- files ---- */
int pak_parse_float(const char *s, float *out);     /* = sscanf(s, "%f", out) > 0, fast path for plain decimals */
struct entries *open_entries(const char *name, int labels_needed, int skip_empty);
int save_entries_wcomments(struct entries *codes, const char *name, const char *comments);
#define save_entries(c, n) save_entries_wcomments((c), (n), NULL)
/* raw fp32 side format ("#!somf32", paklib.c) -- open_entries reads it transparently, this writes it */
int save_entries_f32(struct entries *c, const char *name);
/* `-din gen:...` without a host copy: tools whose whole use of the data is epoch-level calls of the engine set this
 * before open_entries(); the rows are then generated in HBM by somhip_dataset_generate when the mirror is made (a
 * 10 M x 512 stream is 20 GB the host never allocates, parses or sends over PCIe).  Sources with labels=1, and any
 * later need for host rows (-rand, -buffer), materialise as before. */
extern int pak_gen_virtual_ok;
int pak_materialize(struct entries *e);
/* the seeded Gaussian-mixture stream behind `-din gen:k=..,dim=..,n=..,seed=..[,labels=1]` (paklib.c) */
uint64_t pak_splitmix64(uint64_t x);
float pak_gen_z(uint64_t seed, uint64_t counter);
void pak_gen_row(uint64_t seed, int k_centres, int dim, long row, float *out, int *centre);
int pak_parse_gen(const char *spec, long *n, int *dim, int *k, uint64_t *seed, int *labels);
void close_entries(struct entries *);
void clear_entry_labels(struct entries *e, long row);
void add_entry_label(struct entries *e, long row, int label);
int alpha_read(float *alpha, long noc, const char *infile);
int alpha_write(float *alpha, long noc, const char *outfile);
void invalidate_alphafile(const char *outfile);

/* ---- RNG + -rand shuffle (lvq_pak.c:459-484, datafile.c:1152-1188) ---- */
void init_random(int seed);
long orand(void);
void randomize_entry_order(struct entries *e);

/* ---- the registry (-selfuncs) and the epoch-level functions behind it ---- */
int set_teach_params(struct teach_params *p, struct entries *codes, struct entries *data, const char *funcname);
int set_som_params(struct teach_params *p);
ALPHA_FUNC linear_alpha, inverse_t_alpha;
ALPHA_FUNC *alpha_func_by_name(const char *name, short *id);
extern int use_fixed_level, use_weights_level;

struct entries *som_training(struct teach_params *teach);
struct entries *lvq1_training(struct teach_params *teach);
struct entries *olvq1_training(struct teach_params *teach, const char *infile, const char *outfile);
struct entries *lvq2_training(struct teach_params *teach, float winlen);
struct entries *lvq3_training(struct teach_params *teach, float epsilon, float winlen);
float find_qerror(struct teach_params *teach);
float find_qerror2(struct teach_params *teach);   /* qerror -qetype 1 (som_rout.c:823) */
/* winners of every data row (the scan behind compute_accuracy / find_labels) */
int find_all_winners(struct teach_params *teach, int32_t *index, float *diff, int32_t *ret);
/* k-NN consumers (eveninit/propinit, knntest, classify): k <= 8 */
int find_all_knn(struct entries *codes, struct entries *data, int knn, int32_t *index, float *diff);
unsigned char *knn_correct_all(struct entries *data, int knn);
struct entries *pick_rows(struct entries *src, const long *rows, long n);
struct entries *lininit_codes(struct entries *data, int topol, int neigh, int xdim, int ydim);
struct entries *randinit_codes(struct entries *data, int topol, int neigh, int xdim, int ydim);
/* ---- what every tool does before the engine is involved ---- */
struct pak_inputs { struct entries *data, *codes; };
/* opens -din then -cin with the reference's messages; fmt strings take the file name.  need_map:
 * the codebook must be a hexa/rect map.  Returns 0 or 1 (after closing what it opened). */
int pak_open_inputs(const char *din, int data_labels, const char *data_fail_fmt, const char *cin, int code_labels,
                    const char *code_fail_fmt, int need_map, struct pak_inputs *io);
/* options shared by the training tools: -din -cin -cout -rlen -rand -buffer -alpha_type -selfuncs
 * -snapfile -snapinterval (vsom.c:77-131, lvqtrain.c:115-142) */
struct pak_train_cli {
  char *din, *cin, *cout, *rand_s, *alpha_s, *funcname;
  long length, buffer;
  struct snapshot_info snap;
  int want_snapshots;
};
void pak_train_cli(int argc, char **argv, struct pak_train_cli *o);
/* init_random(-rand) and the data order it asks for: one shuffle of the whole file, or per-buffer
 * reshuffling when -buffer is smaller than the file (datafile.c:237-344) */
void pak_apply_rand(struct entries *data, const char *rand_s, long buffer);
void pak_shutdown(void);
/* vsom -gpus G: one process per GPU, codebook sharded, RCCL all-reduce of the winner keys (paklib.c).  Must be called
 * before this process has used a GPU; rank 0 runs after(teach, arg) (save the codebook) when training succeeded. */
int som_training_multi(struct teach_params *teach, int gpus, int (*after)(struct teach_params *, void *), void *arg);
/* lvqtrain -gpus G: rows of the codebook sharded over the ranks, exact (paklib.c) */
int lvq_training_multi(struct teach_params *teach, int kind, float winlen, float epsilon, float clamp, float *talpha, int gpus,
                       int (*after)(struct teach_params *, void *), void *arg);
/* the launcher behind it, for tools that spread independent work over the GPUs (vfind -gpus G: trials as replicas) */
int pak_run_ranks(int world, int (*rank_main)(int rank, int world, int *fds, void *arg), void *arg);
int pak_rank_device(int rank);          /* selects GPU rank % (visible GPUs) for this process's engine; -1 on failure */
int pak_sock_write(int fd, const void *b, size_t n);
int pak_sock_read(int fd, void *b, size_t n);

#endif
