/* som_lvq_oracle.h -- TEST INFRASTRUCTURE ONLY (see som_lvq_oracle.c). */
#ifndef SOM_LVQ_ORACLE_H
#define SOM_LVQ_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ids as the reference defines them (lvq_pak.h:209-224) */
#define ORC_TOPOL_HEXA 3
#define ORC_TOPOL_RECT 4
#define ORC_NEIGH_BUBBLE 1
#define ORC_NEIGH_GAUSSIAN 2
#define ORC_ALPHA_LINEAR 1
#define ORC_ALPHA_INVERSE_T 2
#define ORC_LVQ1 1
#define ORC_OLVQ1 2
#define ORC_LVQ2 3
#define ORC_LVQ3 4

int   orc_find_winner_euc(const float *codes, long n, int d, const float *x,
                          const unsigned char *mask, long *index, float *diff);
int   orc_find_winner_knn(const float *codes, long n, int d, const float *x,
                          const unsigned char *mask, int knn, long *index, float *diff);
float orc_vector_dist_euc(const float *a, const unsigned char *ma, const float *b,
                          const unsigned char *mb, int d);
void  orc_adapt_vector(float *c, const float *x, const unsigned char *mask, int d, float alpha);
float orc_hexa_dist(int bx, int by, int tx, int ty);
float orc_rect_dist(int bx, int by, int tx, int ty);
float orc_mapdist(int topol, int bx, int by, int tx, int ty);
float orc_alpha(int type, long iter, long length, float alpha);
float orc_som_radius(long iter, long length, float radius);
float orc_weighted_alpha(float talp, float weight);
float orc_gaussian_h(float dd, float radius, float alpha);

int orc_som_training(float *codes, long n, int d, int xdim, int ydim, int topol, int neigh,
                     const float *data, long ndata, const short *weight, const short *fixed_xy,
                     const unsigned char *mask, long length, float alpha, float radius,
                     int alpha_type, int fixed_on, int weights_on, long batch,
                     long *trace_index, float *trace_diff);

int orc_lvq_training(int kind, float *codes, const int *clabels, long n, int d,
                     const float *data, const int *dlabels, long ndata, long length,
                     float alpha, int alpha_type, float winlen, float epsilon,
                     float *talpha, long *trace_index, float *trace_diff);

int   orc_winners(const float *codes, long n, int d, const float *data, long ndata,
                  const unsigned char *mask, int knn, int use_knn_fn,
                  long *index, float *diff, int *ret);
float orc_find_qerror(const float *codes, long n, int d, const float *data, long ndata,
                      const unsigned char *mask, long *index, float *diff);
float orc_qerror_from_diffs(const float *diff, const int *ret, long ndata);
float orc_find_qerror2(const float *codes, long n, int d, int xdim, int topol, int neigh,
                       const float *data, long ndata, const unsigned char *mask, float radius);
long  orc_accuracy(const float *codes, const int *clabels, long n, int d, const float *data,
                   const int *dlabels, long ndata, unsigned char *correct);
void  orc_unit_hits(const float *codes, long n, int d, const float *data, const int *dlabels,
                    long ndata, const unsigned char *mask, int nlabels, long *hits);

void  orc_srand(int seed);
long  orc_rand(void);
void  orc_shuffle_perm(long n, int seed, long *perm);
int   orc_randinit(const float *data, long ndata, int d, int xdim, int ydim, int seed,
                   float *codes_out);

#ifdef __cplusplus
}
#endif
#endif
