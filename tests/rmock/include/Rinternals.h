/* Minimal mock of R's C API (TEST INFRASTRUCTURE): just the entry points that
 * shim/sgdnet_shim.c uses, with R's semantics, so that the shim -- the code a
 * maintainer drops into the sgdnet package in place of src/RcppExports.cpp --
 * is compiled and driven end to end on machines without R.  Not a stand-in for
 * building the reference: nothing of /root/reference is compiled against it. */
#ifndef RMOCK_RINTERNALS_H_
#define RMOCK_RINTERNALS_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;   /* R_ext/Boolean.h */
typedef struct rmock_sexprec* SEXP;
typedef unsigned int SEXPTYPE;

#define NILSXP   0
#define SYMSXP   1
#define CHARSXP  9
#define LGLSXP  10
#define INTSXP  13
#define REALSXP 14
#define STRSXP  16
#define VECSXP  19
#define S4SXP   25

#define NA_INTEGER (-2147483647 - 1)
#define NA_LOGICAL NA_INTEGER

extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol, R_GlobalEnv, R_UnboundValue;

int      TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int      LENGTH(SEXP x);
double*  REAL(SEXP x);       /* errors unless REALSXP, like R */
int*     INTEGER(SEXP x);    /* INTSXP or LGLSXP */
int*     LOGICAL(SEXP x);
const char* CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_mkChar(const char* s);
SEXP Rf_mkString(const char* s);
SEXP Rf_ScalarReal(double v);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_ScalarLogical(int v);
SEXP Rf_install(const char* name);
SEXP Rf_getAttrib(SEXP x, SEXP sym);
SEXP Rf_setAttrib(SEXP x, SEXP sym, SEXP val);
SEXP R_do_slot(SEXP obj, SEXP name);
SEXP Rf_coerceVector(SEXP x, SEXPTYPE type);
int    Rf_asLogical(SEXP x);
int    Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
SEXP   Rf_asChar(SEXP x);
int    Rf_isNull(SEXP x);
SEXP   Rf_GetOption1(SEXP tag);
SEXP   Rf_findVar(SEXP sym, SEXP env);          /* R_UnboundValue when absent */
void   Rf_defineVar(SEXP sym, SEXP value, SEXP env);
SEXP   Rf_protect(SEXP x);
void   Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
void   Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
char*  R_alloc(size_t n, int size);

#ifdef __cplusplus
}
#endif
#endif
