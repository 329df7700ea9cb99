#ifndef RMOCK_RANDOM_H_
#define RMOCK_RANDOM_H_
#ifdef __cplusplus
extern "C" {
#endif
void   GetRNGstate(void);
void   PutRNGstate(void);
double unif_rand(void);
#ifdef __cplusplus
}
#endif
#endif
