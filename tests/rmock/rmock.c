/* rmock.c -- a few hundred lines of R's C API (TEST INFRASTRUCTURE, see include/Rinternals.h).
 *
 * Object model: every SEXP is a heap record {type, length, payload, attribute list}; symbols are
 * interned; everything lives in an arena that rmock_reset() frees.  Rf_error() longjmps back to
 * rmock_call3(), the stand-in for .Call().  The random-number side is R's default generator
 * (Mersenne-Twister, set.seed() scrambling and unif_rand() fix-up as in R's src/main/RNG.c) with
 * counters, so a test can check that the shim consumed exactly the draws the reference would.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R_ext/Rdynload.h"
#include "R_ext/Random.h"
#include "Rinternals.h"

struct attr_node {
  SEXP tag, val;
  struct attr_node* next;
};

struct rmock_sexprec {
  int type;
  R_xlen_t len;
  void* data;               /* int[], double[], SEXP[], char[] */
  struct attr_node* attrib;
  struct rmock_sexprec* arena_next;
};

static struct rmock_sexprec nil_rec = {NILSXP, 0, NULL, NULL, NULL};
static struct rmock_sexprec env_rec = {4 /* ENVSXP */, 0, NULL, NULL, NULL};
static struct rmock_sexprec unbound_rec = {SYMSXP, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_GlobalEnv = &env_rec;          /* its variables are kept as the record's attribute list */
SEXP R_UnboundValue = &unbound_rec;
SEXP R_NamesSymbol = NULL, R_DimSymbol = NULL;

static SEXP arena = NULL;
static void* raw_arena[1 << 16];
static int raw_count = 0;
static jmp_buf* err_jmp = NULL;
static char err_msg[1024];
static int protect_depth = 0, protect_max = 0;

/* ---- symbols, options ---- */
static SEXP symbols[512];
static int n_symbols = 0;
static struct attr_node* options = NULL;

static SEXP new_rec(int type, R_xlen_t len, size_t bytes) {
  SEXP x = (SEXP)calloc(1, sizeof(*x));
  x->type = type;
  x->len = len;
  x->data = bytes ? calloc(1, bytes) : NULL;
  x->arena_next = arena;
  arena = x;
  return x;
}

SEXP Rf_install(const char* name) {
  for (int i = 0; i < n_symbols; ++i)
    if (strcmp((const char*)symbols[i]->data, name) == 0) return symbols[i];
  SEXP s = (SEXP)calloc(1, sizeof(*s));     /* symbols survive rmock_reset() */
  s->type = SYMSXP;
  s->len = (R_xlen_t)strlen(name);
  s->data = strdup(name);
  symbols[n_symbols++] = s;
  return s;
}

static void init_once(void) {
  if (!R_NamesSymbol) {
    R_NamesSymbol = Rf_install("names");
    R_DimSymbol = Rf_install("dim");
  }
}

void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_msg, sizeof(err_msg), fmt, ap);
  va_end(ap);
  if (err_jmp) longjmp(*err_jmp, 1);
  fprintf(stderr, "rmock: Rf_error outside rmock_call3: %s\n", err_msg);
  abort();
}

int TYPEOF(SEXP x) { return x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
int LENGTH(SEXP x) { return (int)x->len; }

double* REAL(SEXP x) {
  if (x->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a '%d'", x->type);
  return (double*)x->data;
}
int* INTEGER(SEXP x) {
  if (x->type != INTSXP && x->type != LGLSXP)
    Rf_error("INTEGER() can only be applied to a 'integer', not a '%d'", x->type);
  return (int*)x->data;
}
int* LOGICAL(SEXP x) {
  if (x->type != LGLSXP) Rf_error("LOGICAL() can only be applied to a 'logical', not a '%d'", x->type);
  return (int*)x->data;
}
const char* CHAR(SEXP x) {
  if (x->type != CHARSXP) Rf_error("CHAR() can only be applied to a 'CHARSXP', not a '%d'", x->type);
  return (const char*)x->data;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i) {
  if (x->type != STRSXP) Rf_error("STRING_ELT() can only be applied to a 'character vector', not a '%d'", x->type);
  if (i < 0 || i >= x->len) Rf_error("attempt to access index %ld/%ld in STRING_ELT", (long)i, (long)x->len);
  return ((SEXP*)x->data)[i];
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) {
  if (x->type != VECSXP) Rf_error("VECTOR_ELT() can only be applied to a 'list', not a '%d'", x->type);
  if (i < 0 || i >= x->len) Rf_error("attempt to access index %ld/%ld in VECTOR_ELT", (long)i, (long)x->len);
  return ((SEXP*)x->data)[i];
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) {
  if (x->type != STRSXP || v->type != CHARSXP || i < 0 || i >= x->len) Rf_error("bad SET_STRING_ELT");
  ((SEXP*)x->data)[i] = v;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
  if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("bad SET_VECTOR_ELT");
  ((SEXP*)x->data)[i] = v;
  return v;
}

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) {
  init_once();
  if (n < 0) Rf_error("negative length vectors are not allowed");
  switch (type) {
    case LGLSXP:
    case INTSXP: return new_rec((int)type, n, sizeof(int) * (size_t)(n ? n : 1));
    case REALSXP: return new_rec(REALSXP, n, sizeof(double) * (size_t)(n ? n : 1));
    case STRSXP:
    case VECSXP: {
      SEXP x = new_rec((int)type, n, sizeof(SEXP) * (size_t)(n ? n : 1));
      for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)x->data)[i] = type == VECSXP ? R_NilValue : Rf_mkChar("");
      return x;
    }
    case S4SXP: return new_rec(S4SXP, 0, 0);
    default: Rf_error("rmock: allocVector of type %u is not mocked", type);
  }
}

SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
  SEXP x = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
  SEXP dim = Rf_allocVector(INTSXP, 2);
  INTEGER(dim)[0] = nrow;
  INTEGER(dim)[1] = ncol;
  Rf_setAttrib(x, R_DimSymbol, dim);
  return x;
}

SEXP Rf_mkChar(const char* s) {
  SEXP x = new_rec(CHARSXP, (R_xlen_t)strlen(s), strlen(s) + 1);
  memcpy(x->data, s, strlen(s) + 1);
  return x;
}
SEXP Rf_mkString(const char* s) {
  SEXP x = Rf_allocVector(STRSXP, 1);
  SET_STRING_ELT(x, 0, Rf_mkChar(s));
  return x;
}
SEXP Rf_ScalarReal(double v) {
  SEXP x = Rf_allocVector(REALSXP, 1);
  REAL(x)[0] = v;
  return x;
}
SEXP Rf_ScalarInteger(int v) {
  SEXP x = Rf_allocVector(INTSXP, 1);
  INTEGER(x)[0] = v;
  return x;
}
SEXP Rf_ScalarLogical(int v) {
  SEXP x = Rf_allocVector(LGLSXP, 1);
  LOGICAL(x)[0] = v;
  return x;
}

SEXP Rf_getAttrib(SEXP x, SEXP sym) {
  for (struct attr_node* a = x->attrib; a; a = a->next)
    if (a->tag == sym) return a->val;
  return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP sym, SEXP val) {
  for (struct attr_node* a = x->attrib; a; a = a->next)
    if (a->tag == sym) {
      a->val = val;
      return val;
    }
  struct attr_node* a = (struct attr_node*)calloc(1, sizeof(*a));
  if (raw_count < (int)(sizeof(raw_arena) / sizeof(raw_arena[0]))) raw_arena[raw_count++] = a;
  a->tag = sym;
  a->val = val;
  a->next = x->attrib;
  x->attrib = a;
  return val;
}
SEXP R_do_slot(SEXP obj, SEXP name) {
  SEXP v = Rf_getAttrib(obj, name);
  if (v == R_NilValue) Rf_error("no slot of name \"%s\" for this object", (const char*)name->data);
  return v;
}

SEXP Rf_coerceVector(SEXP x, SEXPTYPE type) {
  if ((SEXPTYPE)x->type == type) return x;
  SEXP y = Rf_allocVector(type, x->len);
  y->attrib = x->attrib;                           /* dim, names carried over like R does */
  for (R_xlen_t i = 0; i < x->len; ++i) {
    double v;
    int na = 0;
    if (x->type == REALSXP) v = ((double*)x->data)[i], na = v != v;
    else if (x->type == INTSXP || x->type == LGLSXP) na = ((int*)x->data)[i] == NA_INTEGER, v = ((int*)x->data)[i];
    else Rf_error("rmock: coerceVector from type %d is not mocked", x->type);
    if (type == REALSXP) ((double*)y->data)[i] = na ? (0.0 / 0.0) : v;
    else if (type == INTSXP) ((int*)y->data)[i] = na ? NA_INTEGER : (int)v;
    else if (type == LGLSXP) ((int*)y->data)[i] = na ? NA_LOGICAL : (v != 0.0);
    else Rf_error("rmock: coerceVector to type %u is not mocked", type);
  }
  return y;
}

static double as_double(SEXP x, int* na) {
  *na = 0;
  if (x->len < 1) {
    *na = 1;
    return 0.0;
  }
  if (x->type == REALSXP) {
    const double v = ((double*)x->data)[0];
    *na = v != v;
    return v;
  }
  if (x->type == INTSXP || x->type == LGLSXP) {
    const int v = ((int*)x->data)[0];
    *na = v == NA_INTEGER;
    return (double)v;
  }
  *na = 1;
  return 0.0;
}
int Rf_asLogical(SEXP x) {
  int na;
  const double v = as_double(x, &na);
  return na ? NA_LOGICAL : (v != 0.0);
}
int Rf_asInteger(SEXP x) {
  int na;
  const double v = as_double(x, &na);
  if (na || v >= 2147483648.0 || v <= -2147483649.0) return NA_INTEGER;
  return (int)v;
}
double Rf_asReal(SEXP x) {
  int na;
  const double v = as_double(x, &na);
  return na ? (0.0 / 0.0) : v;
}
SEXP Rf_asChar(SEXP x) {
  if (x->type == STRSXP && x->len >= 1) return ((SEXP*)x->data)[0];
  if (x->type == CHARSXP) return x;
  return Rf_mkChar("NA");
}
int Rf_isNull(SEXP x) { return x == R_NilValue; }

SEXP Rf_GetOption1(SEXP tag) {
  for (struct attr_node* a = options; a; a = a->next)
    if (a->tag == tag) return a->val;
  return R_NilValue;
}

SEXP Rf_findVar(SEXP sym, SEXP env) {
  for (struct attr_node* a = env->attrib; a; a = a->next)
    if (a->tag == sym) return a->val;
  return R_UnboundValue;
}
void Rf_defineVar(SEXP sym, SEXP value, SEXP env) { Rf_setAttrib(env, sym, value); }

SEXP Rf_protect(SEXP x) {
  if (++protect_depth > protect_max) protect_max = protect_depth;
  if (protect_depth > 10000) Rf_error("protect(): protection stack overflow");
  return x;
}
void Rf_unprotect(int n) {
  if (n > protect_depth) Rf_error("unprotect(): only %d protected items", protect_depth);
  protect_depth -= n;
}

char* R_alloc(size_t n, int size) {
  void* q = calloc(n ? n : 1, (size_t)size);
  if (!q) Rf_error("cannot allocate memory block of size %.1f Gb", (double)n * size / 1073741824.0);
  if (raw_count >= (int)(sizeof(raw_arena) / sizeof(raw_arena[0]))) Rf_error("rmock: raw arena full");
  raw_arena[raw_count++] = q;
  return (char*)q;
}

/* ---- R's default RNG ---- */
static uint32_t mt[624];
static int mti = 625;
static long long n_unif = 0;
static int n_get = 0, n_put = 0;
void rmock_set_seed(uint32_t seed);
void PutRNGstate(void);

void rmock_set_seed(uint32_t seed) {               /* set.seed(seed) */
  for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
  uint32_t i_seed[625];
  for (int j = 0; j < 625; ++j) {
    seed = 69069u * seed + 1u;
    i_seed[j] = seed;
  }
  memcpy(mt, i_seed + 1, sizeof(mt));
  mti = 624;                                       /* FixupSeeds: dummy[0] = 624 */
  PutRNGstate();                                   /* set.seed() leaves .Random.seed in the global environment */
  n_unif = 0;
  n_get = n_put = 0;
}

double unif_rand(void) {
  static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
  if (mti >= 624) {
    if (mti == 625) rmock_set_seed(4357u);
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; ++kk) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < 623; ++kk) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
    mti = 0;
  }
  uint32_t y = mt[mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  ++n_unif;
  double v = (double)y * 2.3283064365386963e-10;   /* [0,1) */
  if (v <= 0.0) return 0.5 * 2.328306437080797e-10;
  if (1.0 - v <= 0.0) return 1.0 - 0.5 * 2.328306437080797e-10;
  return v;
}
/* .Random.seed: integer(626) = { kind code, mti, mt[624] }; code = RNG kind (3 = Mersenne-Twister)
 * + 100 * normal kind (3 = Inversion) + 10000 * sample kind (R >= 3.6: 1 = Rejection) */
static int rng_kind_code = 10403;
void GetRNGstate(void) {
  ++n_get;
  SEXP seed = Rf_findVar(Rf_install(".Random.seed"), R_GlobalEnv);
  if (seed == R_UnboundValue) {
    if (mti == 625) rmock_set_seed(4357u);        /* R would randomise from the clock */
    return;
  }
  if (seed->type != INTSXP || seed->len != 626) Rf_error("'.Random.seed' has wrong length");
  const int* v = (const int*)seed->data;
  rng_kind_code = v[0];
  mti = v[1];
  memcpy(mt, v + 2, sizeof(mt));
}
void PutRNGstate(void) {
  ++n_put;
  SEXP seed = Rf_allocVector(INTSXP, 626);
  int* v = (int*)seed->data;
  v[0] = rng_kind_code;
  v[1] = mti;
  memcpy(v + 2, mt, sizeof(mt));
  Rf_defineVar(Rf_install(".Random.seed"), seed, R_GlobalEnv);
}
void rmock_set_rng_kind(int code) { rng_kind_code = code; }

/* ---- registration ---- */
static const R_CallMethodDef* registered = NULL;
static int dynamic_symbols = -1;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* c, const R_CallMethodDef* call,
                       const R_FortranMethodDef* f, const R_ExternalMethodDef* e) {
  (void)info; (void)c; (void)f; (void)e;
  registered = call;
  return 1;
}
int R_useDynamicSymbols(DllInfo* info, int value) {
  (void)info;
  dynamic_symbols = value;
  return 1;
}

/* ================= test-side helpers (called from Python through ctypes) ================= */
void rmock_reset(void) {
  env_rec.attrib = NULL;
  rng_kind_code = 10403;
  while (arena) {
    SEXP nx = arena->arena_next;
    free(arena->data);
    free(arena);
    arena = nx;
  }
  for (int i = 0; i < raw_count; ++i) free(raw_arena[i]);
  raw_count = 0;
  options = NULL;
  protect_depth = protect_max = 0;
  registered = NULL;
  dynamic_symbols = -1;
}
int rmock_protect_depth(void) { return protect_depth; }
long long rmock_unif_count(void) { return n_unif; }
int rmock_rng_scope_calls(void) { return n_get * 100 + n_put; }
void rmock_rng_state(uint32_t* out625) {           /* .Random.seed[2:626]: mti, mt[624] */
  out625[0] = (uint32_t)mti;
  memcpy(out625 + 1, mt, sizeof(mt));
}
const char* rmock_last_error(void) { return err_msg; }
int rmock_registered(int i, char* name, int cap, int* nargs) {
  if (!registered || !registered[i].name) return 0;
  snprintf(name, (size_t)cap, "%s", registered[i].name);
  *nargs = registered[i].numArgs;
  return 1;
}
void* rmock_registered_fn(const char* name) {
  for (int i = 0; registered && registered[i].name; ++i)
    if (strcmp(registered[i].name, name) == 0) return (void*)registered[i].fun;
  return NULL;
}
int rmock_dynamic_symbols(void) { return dynamic_symbols; }

void rmock_set_option(const char* name, SEXP val) {
  struct attr_node* a = (struct attr_node*)calloc(1, sizeof(*a));
  raw_arena[raw_count++] = a;
  a->tag = Rf_install(name);
  a->val = val;
  a->next = options;
  options = a;
}
SEXP rmock_real(const double* v, R_xlen_t n) {
  SEXP x = Rf_allocVector(REALSXP, n);
  if (n) memcpy(x->data, v, sizeof(double) * (size_t)n);
  return x;
}
SEXP rmock_int(const int* v, R_xlen_t n, int logical) {
  SEXP x = Rf_allocVector(logical ? LGLSXP : INTSXP, n);
  if (n) memcpy(x->data, v, sizeof(int) * (size_t)n);
  return x;
}
SEXP rmock_str(const char* s) { return Rf_mkString(s); }
SEXP rmock_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
void rmock_list_set(SEXP list, R_xlen_t i, const char* name, SEXP val) {
  init_once();
  SET_VECTOR_ELT(list, i, val);
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  if (names == R_NilValue) {
    names = Rf_allocVector(STRSXP, list->len);
    Rf_setAttrib(list, R_NamesSymbol, names);
  }
  SET_STRING_ELT(names, i, Rf_mkChar(name));
}
void rmock_set_dim(SEXP x, int nrow, int ncol) {
  init_once();
  SEXP dim = Rf_allocVector(INTSXP, 2);
  INTEGER(dim)[0] = nrow;
  INTEGER(dim)[1] = ncol;
  Rf_setAttrib(x, R_DimSymbol, dim);
}
SEXP rmock_s4(void) { return Rf_allocVector(S4SXP, 0); }
void rmock_set_slot(SEXP obj, const char* name, SEXP val) { Rf_setAttrib(obj, Rf_install(name), val); }

/* .Call(fn, x, y, control): returns NULL and leaves the message in rmock_last_error() if the
 * callee raised an R error */
typedef SEXP (*call3_fn)(SEXP, SEXP, SEXP);
SEXP rmock_call3(void* fn, SEXP a, SEXP b, SEXP c) {
  jmp_buf jb;
  err_jmp = &jb;
  err_msg[0] = 0;
  SEXP out = NULL;
  const int depth = protect_depth;
  if (setjmp(jb) == 0) out = ((call3_fn)fn)(a, b, c);
  else protect_depth = depth;                      /* R unwinds the protection stack on error */
  err_jmp = NULL;
  return out;
}

/* readers for the returned object */
int rmock_typeof(SEXP x) { return x->type; }
long long rmock_length(SEXP x) { return (long long)x->len; }
const double* rmock_real_ptr(SEXP x) { return x->type == REALSXP ? (const double*)x->data : NULL; }
SEXP rmock_elt(SEXP x, long long i) { return ((SEXP*)x->data)[i]; }
const char* rmock_name(SEXP list, long long i) {
  init_once();
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  return names == R_NilValue ? "" : (const char*)(((SEXP*)names->data)[i])->data;
}
int rmock_dim(SEXP x, int which) {
  init_once();
  SEXP d = Rf_getAttrib(x, R_DimSymbol);
  return d == R_NilValue ? -1 : ((int*)d->data)[which];
}
