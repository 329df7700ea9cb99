/*
 * sgdnet_shim.c -- the .Call shim that replaces the reference's generated
 * src/RcppExports.cpp (lines 9-45 there) in the sgdnet R package.
 *
 * It exports exactly the three symbols the R front-end expects
 * (R/RcppExports.R:68-75, NAMESPACE:26 `useDynLib(sgdnet, .registration = TRUE)`):
 *
 *     _sgdnet_SgdnetDense(x, y, control)
 *     _sgdnet_SgdnetSparse(x, y, control)
 *     R_init_sgdnet(DllInfo*)
 *
 * and forwards to the C ABI of libsgdnet_hip.so (include/sgdnet_hip.h).  Nothing
 * else of the reference's src/ is needed; R/ stays untouched.
 *
 * Build on a machine that has R.  Neither this container nor the GPU box has
 * R, so here the file is compiled against tests/rmock (a small mock of the R C API) by build.sh
 * and driven end to end by tests/test_shim_mock.py / tests/test_gpu_shim.py:
 *
 *     R CMD SHLIB -o sgdnet.so sgdnet_shim.c -I<repo>/include \
 *         -L<repo>/sgdnet_amd/lib -lsgdnet_hip -Wl,-rpath,<repo>/sgdnet_amd/lib
 *
 * Semantics kept from the reference:
 *   - control fields are looked up BY NAME (src/sgdnet.cpp:76-78,129-138,305,319,326);
 *   - the sample order comes from R's own RNG between GetRNGstate()/PutRNGstate()
 *     (Rcpp::RNGScope, src/RcppExports.cpp:14,27): one unif_rand() per inner
 *     iteration (src/saga-sparse.h:261), so set.seed() reproducibility and the
 *     RNG state after the call are those of the reference;
 *   - the returned list has the reference's names, order and shapes
 *     (src/sgdnet.cpp:275-284);
 *   - inputs are borrowed read-only; failures become R errors after cleanup.
 * Backend extensions are read from R options so that the R code needs no change:
 *   options(sgdnet.mode = "exact" | "batched" | "auto", sgdnet.batch = <int>, sgdnet.device = <int>,
 *           sgdnet.gpus = <int>)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Random.h>
#include <R_ext/Rdynload.h>
#include <stdlib.h>
#include <string.h>

#include "sgdnet_hip.h"

static SEXP list_get(SEXP list, const char* name) {
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  for (R_xlen_t i = 0; i < XLENGTH(list); ++i)
    if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
  Rf_error("control list lacks field '%s'", name);
  return R_NilValue;
}

static int family_code(const char* f) {
  if (strcmp(f, "gaussian") == 0) return SGDNET_GAUSSIAN;
  if (strcmp(f, "binomial") == 0) return SGDNET_BINOMIAL;
  if (strcmp(f, "multinomial") == 0) return SGDNET_MULTINOMIAL;
  if (strcmp(f, "mgaussian") == 0) return SGDNET_MGAUSSIAN;
  Rf_error("unknown family '%s'", f);
  return -1;
}

static double r_unif(void* ctx) { (void)ctx; return unif_rand(); }

/* Sample order without one host call per draw.  With R's default generator (Mersenne-Twister) the
 * backend can continue R's own stream on the device: .Random.seed = { kind code, mti, mt[624] } is
 * copied into a sgdnet_rng (control.rng_state), the fit draws from it bit for bit as unif_rand()
 * would -- 10M draws per epoch in ~1 ms instead of ~50 ms of callbacks -- and the advanced state is
 * written back, so set.seed() reproducibility and the generator state after the call are the
 * reference's.  Any other RNGkind(), or options(sgdnet.rng = "callback"), keeps unif_rand(). */
static int rng_handoff(sgdnet_rng* st) {
  SEXP opt = Rf_GetOption1(Rf_install("sgdnet.rng"));
  if (opt != R_NilValue && strcmp(CHAR(Rf_asChar(opt)), "callback") == 0) return 0;
  PutRNGstate();                              /* R's internal state -> .Random.seed */
  SEXP seed = Rf_findVar(Rf_install(".Random.seed"), R_GlobalEnv);
  if (seed == R_UnboundValue || TYPEOF(seed) != INTSXP || XLENGTH(seed) != 626) return 0;
  const int* v = INTEGER(seed);
  if (v[0] % 100 != 3) return 0;              /* not Mersenne-Twister */
  st->mti = (uint32_t)v[1];
  memcpy(st->mt, v + 2, sizeof(st->mt));
  return 1;
}

static void rng_handback(const sgdnet_rng* st) {
  SEXP old = Rf_findVar(Rf_install(".Random.seed"), R_GlobalEnv);
  SEXP seed = PROTECT(Rf_allocVector(INTSXP, 626));
  int* v = INTEGER(seed);
  v[0] = INTEGER(old)[0];
  v[1] = (int)st->mti;
  memcpy(v + 2, st->mt, sizeof(st->mt));
  Rf_defineVar(Rf_install(".Random.seed"), seed, R_GlobalEnv);
  UNPROTECT(1);
  GetRNGstate();                              /* .Random.seed -> R's internal state */
}

static void fill_control(SEXP control, sgdnet_control* c) {
  memset(c, 0, sizeof(*c));
  c->debug = Rf_asLogical(list_get(control, "debug"));
  c->elasticnet_mix = Rf_asReal(list_get(control, "elasticnet_mix"));
  c->family = family_code(CHAR(Rf_asChar(list_get(control, "family"))));
  c->intercept = Rf_asLogical(list_get(control, "intercept"));
  c->is_sparse = Rf_asLogical(list_get(control, "is_sparse"));
  SEXP lambda = list_get(control, "lambda");
  c->n_lambda_user = (int)XLENGTH(lambda);
  c->lambda = c->n_lambda_user ? REAL(lambda) : NULL;
  c->lambda_min_ratio = Rf_asReal(list_get(control, "lambda_min_ratio"));
  c->max_iter = (unsigned)Rf_asInteger(list_get(control, "max_iter"));
  c->n_lambda = Rf_asInteger(list_get(control, "n_lambda"));
  c->n_classes = Rf_asInteger(list_get(control, "n_classes"));
  c->standardize = Rf_asLogical(list_get(control, "standardize"));
  c->standardize_response = Rf_asLogical(list_get(control, "standardize_response"));
  c->tol = Rf_asReal(list_get(control, "tol"));
  c->type_multinomial =
      strcmp(CHAR(Rf_asChar(list_get(control, "type_multinomial"))), "grouped") == 0;
  c->unif = r_unif;                    /* R's RNG, whatever RNGkind() is active; see rng_handoff() */
  SEXP opt = Rf_GetOption1(Rf_install("sgdnet.mode"));
  if (opt != R_NilValue && strcmp(CHAR(Rf_asChar(opt)), "batched") == 0) c->mode = SGDNET_MODE_BATCHED;
  if (opt != R_NilValue && strcmp(CHAR(Rf_asChar(opt)), "auto") == 0) c->mode = SGDNET_MODE_AUTO;
  opt = Rf_GetOption1(Rf_install("sgdnet.batch"));
  if (opt != R_NilValue) c->batch = (int64_t)Rf_asReal(opt);
  opt = Rf_GetOption1(Rf_install("sgdnet.device"));
  if (opt != R_NilValue) c->device = Rf_asInteger(opt);
  /* options(sgdnet.gpus = N): the fit sharded over GPUs device .. device + N - 1 of this node (ABI 4; batched
   * iteration of sparse x with one response, the backend says so otherwise) */
  opt = Rf_GetOption1(Rf_install("sgdnet.gpus"));
  if (opt != R_NilValue && Rf_asInteger(opt) > 1) c->n_gpus = Rf_asInteger(opt);
}

/* Debug losses (options(sgdnet.debug = TRUE)): the backend reports each lambda's per-epoch losses
 * through control.losses_sink when that lambda is done; they are kept in malloc'ed blocks sized by
 * the epochs actually run (the reference grows a std::vector, src/saga-sparse.h:364 -- reserving
 * n_lambda x max_iter would be 80 GB at the reference's benchmark setting maxit = 1e8).  No R API
 * is called from the sink: an R error there would longjmp through the library's C++ frames. */
typedef struct {
  int      n_lambda;
  double** values;
  int*     count;
  int      failed;
} loss_store;

static void loss_sink(void* ctx, int li, const double* losses, int count) {
  loss_store* st = (loss_store*)ctx;
  if (li < 0 || li >= st->n_lambda) return;
  free(st->values[li]);                       /* a lambda that was run again replaces its record */
  st->values[li] = (double*)malloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
  if (!st->values[li]) {
    st->failed = 1;
    st->count[li] = 0;
    return;
  }
  memcpy(st->values[li], losses, sizeof(double) * (size_t)count);
  st->count[li] = count;
}

static void loss_store_free(loss_store* st) {
  if (!st->values) return;
  for (int i = 0; i < st->n_lambda; ++i) free(st->values[i]);
  free(st->values);
  free(st->count);
  st->values = NULL;
}

/* builds the list of src/sgdnet.cpp:275-284 from a filled sgdnet_result */
static SEXP wrap_result(const sgdnet_control* c, const sgdnet_result* r, const loss_store* ls, int K, R_xlen_t p) {
  const int nl = c->n_lambda;
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 8));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, 8));
  const char* nm[8] = {"a0", "beta", "losses", "npasses", "nulldev", "dev.ratio", "lambda",
                       "return_codes"};
  for (int i = 0; i < 8; ++i) SET_STRING_ELT(names, i, Rf_mkChar(nm[i]));
  Rf_setAttrib(out, R_NamesSymbol, names);

  SEXP a0 = PROTECT(Rf_allocVector(VECSXP, nl));
  SEXP beta = PROTECT(Rf_allocVector(VECSXP, nl));
  for (int i = 0; i < nl; ++i) {
    SEXP a = PROTECT(Rf_allocVector(REALSXP, K));
    memcpy(REAL(a), r->a0 + (size_t)i * K, sizeof(double) * K);
    SET_VECTOR_ELT(a0, i, a);
    SEXP b = PROTECT(Rf_allocMatrix(REALSXP, K, (int)p));
    memcpy(REAL(b), r->beta + (size_t)i * K * p, sizeof(double) * (size_t)K * p);
    SET_VECTOR_ELT(beta, i, b);
    UNPROTECT(2);
  }
  SEXP losses = PROTECT(Rf_allocVector(VECSXP, c->debug ? nl : 0));
  if (c->debug)
    for (int i = 0; i < nl; ++i) {
      SEXP l = PROTECT(Rf_allocVector(REALSXP, ls->count[i]));
      if (ls->count[i]) memcpy(REAL(l), ls->values[i], sizeof(double) * (size_t)ls->count[i]);
      SET_VECTOR_ELT(losses, i, l);
      UNPROTECT(1);
    }
  SEXP dev = PROTECT(Rf_allocVector(REALSXP, nl));
  SEXP lam = PROTECT(Rf_allocVector(REALSXP, nl));
  SEXP rc = PROTECT(Rf_allocVector(REALSXP, nl));
  memcpy(REAL(dev), r->dev_ratio, sizeof(double) * nl);
  memcpy(REAL(lam), r->lambda, sizeof(double) * nl);
  memcpy(REAL(rc), r->return_codes, sizeof(double) * nl);
  SET_VECTOR_ELT(out, 0, a0);
  SET_VECTOR_ELT(out, 1, beta);
  SET_VECTOR_ELT(out, 2, losses);
  SET_VECTOR_ELT(out, 3, Rf_ScalarReal(r->npasses));
  SET_VECTOR_ELT(out, 4, Rf_ScalarReal(r->nulldev));
  SET_VECTOR_ELT(out, 5, dev);
  SET_VECTOR_ELT(out, 6, lam);
  SET_VECTOR_ELT(out, 7, rc);
  UNPROTECT(8);
  return out;
}

static SEXP run_fit(SEXP x, SEXP y, SEXP control, int sparse) {
  sgdnet_control c;
  fill_control(control, &c);
  int nprot = 0;
  /* Rcpp::as<Eigen::MatrixXd> coerces silently (integer matrices, integer responses): do the same */
  y = PROTECT(Rf_coerceVector(y, REALSXP));
  ++nprot;
  R_xlen_t n, p;
  sgdnet_csc csc;
  memset(&csc, 0, sizeof(csc));
  if (sparse) {                                   /* dgCMatrix slots, R/sgdnet.R:226 */
    SEXP dim = R_do_slot(x, Rf_install("Dim"));
    n = INTEGER(dim)[0];
    p = INTEGER(dim)[1];
    csc.n_rows = n;
    csc.n_cols = p;
    csc.colptr = INTEGER(R_do_slot(x, Rf_install("p")));
    csc.rowidx = INTEGER(R_do_slot(x, Rf_install("i")));
    csc.values = REAL(R_do_slot(x, Rf_install("x")));
  } else {
    SEXP dim = Rf_getAttrib(x, R_DimSymbol);
    if (dim == R_NilValue || XLENGTH(dim) != 2) Rf_error("x must be a matrix");
    n = INTEGER(dim)[0];
    p = INTEGER(dim)[1];
    x = PROTECT(Rf_coerceVector(x, REALSXP));
    ++nprot;
  }
  SEXP ydim = Rf_getAttrib(y, R_DimSymbol);       /* as.matrix(y) in R/sgdnet.R:344; a bare vector is n x 1 */
  const int y_cols = (ydim == R_NilValue || XLENGTH(ydim) != 2) ? 1 : INTEGER(ydim)[1];
  if (XLENGTH(y) != (R_xlen_t)n * y_cols) Rf_error("the number of samples in 'x' and 'y' must match");
  const int K = c.n_classes, nl = c.n_lambda;
  if (K <= 0 || nl <= 0) Rf_error("control$n_classes and control$n_lambda must be positive");
  sgdnet_result r;
  memset(&r, 0, sizeof(r));
  /* R_alloc memory is reclaimed by R at the end of .Call, also on error */
  r.a0 = (double*)R_alloc((size_t)K * nl, sizeof(double));
  r.beta = (double*)R_alloc((size_t)K * p * nl, sizeof(double));
  r.lambda = (double*)R_alloc(nl, sizeof(double));
  r.dev_ratio = (double*)R_alloc(nl, sizeof(double));
  r.return_codes = (double*)R_alloc(nl, sizeof(double));
  loss_store ls;
  memset(&ls, 0, sizeof(ls));
  if (c.debug) {
    ls.n_lambda = nl;
    ls.values = (double**)calloc((size_t)nl, sizeof(double*));
    ls.count = (int*)calloc((size_t)nl, sizeof(int));
    if (!ls.values || !ls.count) {
      free(ls.values);
      free(ls.count);
      Rf_error("cannot allocate the debug-loss table");
    }
    c.losses_sink = loss_sink;
    c.losses_ctx = &ls;
  }
  GetRNGstate();                                  /* Rcpp::RNGScope */
  sgdnet_rng rstate;
  const int handed = rng_handoff(&rstate);
  if (handed) {
    c.unif = NULL;
    c.rng_state = &rstate;
  }
  int rc = sparse ? sgdnet_fit_sparse(&csc, REAL(y), y_cols, &c, &r)
                  : sgdnet_fit_dense(REAL(x), n, p, REAL(y), y_cols, &c, &r);
  if (handed && rc == SGDNET_OK) rng_handback(&rstate);
  PutRNGstate();
  if (rc != SGDNET_OK || ls.failed) {
    loss_store_free(&ls);
    Rf_error("sgdnet (HIP backend): %s", rc != SGDNET_OK ? sgdnet_last_error() : "out of memory for debug losses");
  }
  /* (an R allocation failure inside wrap_result would leak the malloc'ed loss blocks; everything
   * else is R_alloc/PROTECT memory that R reclaims) */
  SEXP out = PROTECT(wrap_result(&c, &r, &ls, K, p));
  ++nprot;
  loss_store_free(&ls);
  UNPROTECT(nprot);
  return out;
}

SEXP _sgdnet_SgdnetDense(SEXP x, SEXP y, SEXP control) { return run_fit(x, y, control, 0); }
SEXP _sgdnet_SgdnetSparse(SEXP x, SEXP y, SEXP control) { return run_fit(x, y, control, 1); }

static const R_CallMethodDef CallEntries[] = {
    {"_sgdnet_SgdnetDense", (DL_FUNC)&_sgdnet_SgdnetDense, 3},
    {"_sgdnet_SgdnetSparse", (DL_FUNC)&_sgdnet_SgdnetSparse, 3},
    {NULL, NULL, 0}};

void R_init_sgdnet(DllInfo* dll) {
  /* the structs this file fills are laid out for SGDNET_ABI_VERSION of the header it was compiled against */
  if (sgdnet_abi_version() != SGDNET_ABI_VERSION)
    Rf_error("libsgdnet_hip has ABI version %d, this shim was built for %d: rebuild the package", sgdnet_abi_version(),
             SGDNET_ABI_VERSION);
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
