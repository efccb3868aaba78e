/* tests/rapi_stub/rapi_stub.c -- implementation of the stand-in declared in Rinternals.h and the R_ext headers of this directory.
 * TEST INFRASTRUCTURE (see Rinternals.h): behaviour as documented in "Writing R Extensions"; written for this repository. */
#include "Rinternals.h"
#include "R_ext/Rdynload.h"
#include "R_ext/Utils.h"
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static struct rapi_sexprec nil_rec = {NILSXP, 0, NULL, NULL, NULL};
static struct rapi_sexprec names_sym = {CHARSXP, 5, NULL, NULL, (void*)"names"};
static struct rapi_sexprec dim_sym = {CHARSXP, 3, NULL, NULL, (void*)"dim"};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol = &names_sym;
SEXP R_DimSymbol = &dim_sym;
double R_PosInf = INFINITY, R_NegInf = -INFINITY, R_NaReal = NAN;
char rapi_last_error[4096] = "";
int rapi_interrupt_checks = 0;

/* every record / transient block is remembered so that a test can release them (and ASan sees no leak of ours) */
typedef struct blk { struct blk* next; void* p; } blk;
static blk* all_recs = NULL;
static blk* transient = NULL;
static int protect_depth = 0;
static jmp_buf* top = NULL;

static void die(const char* what) {
  fprintf(stderr, "rapi_stub: %s\n", what);
  abort();
}
static void* keep(blk** list, void* p) {
  blk* b = (blk*)malloc(sizeof(blk));
  if (!b || !p) die("out of memory");
  b->p = p; b->next = *list; *list = b;
  return p;
}
static void free_list(blk** list) {
  while (*list) { blk* b = *list; *list = b->next; free(b->p); free(b); }
}

static size_t elt_size(SEXPTYPE t) {
  switch (t) {
    case REALSXP: return sizeof(double);
    case INTSXP: case LGLSXP: return sizeof(int);
    case STRSXP: case VECSXP: return sizeof(SEXP);
    case CHARSXP: return 1;
    default: die("allocVector: unsupported type"); return 0;
  }
}
SEXP allocVector(SEXPTYPE type, R_xlen_t n) {
  if (n < 0) die("allocVector: negative length");
  SEXP x = (SEXP)keep(&all_recs, calloc(1, sizeof(struct rapi_sexprec)));
  x->type = type; x->length = n;
  const size_t bytes = (size_t)(n + (type == CHARSXP ? 1 : 0)) * elt_size(type);
  x->data = keep(&all_recs, malloc(bytes ? bytes : 1));
  if (type == STRSXP || type == VECSXP)          /* R initialises these (to "" / NULL); numeric payloads stay uninitialised */
    for (R_xlen_t i = 0; i < n; i++) ((SEXP*)x->data)[i] = R_NilValue;
  else if (type != CHARSXP)
    memset(x->data, 0xA5, bytes);                  /* poison: a shim that forgets to fill a result shows up */
  return x;
}
static SEXP with_dim(SEXP x, int n, const int* d) {
  SEXP dim = allocVector(INTSXP, n);
  for (int i = 0; i < n; i++) ((int*)dim->data)[i] = d[i];
  x->dim = dim;
  return x;
}
SEXP allocMatrix(SEXPTYPE type, int nrow, int ncol) {
  if (nrow < 0 || ncol < 0) die("allocMatrix: negative extent");
  const int d[2] = {nrow, ncol};
  return with_dim(allocVector(type, (R_xlen_t)nrow * ncol), 2, d);
}
SEXP alloc3DArray(SEXPTYPE type, int nrow, int ncol, int nface) {
  if (nrow < 0 || ncol < 0 || nface < 0) die("alloc3DArray: negative extent");
  const int d[3] = {nrow, ncol, nface};
  return with_dim(allocVector(type, (R_xlen_t)nrow * ncol * nface), 3, d);
}
SEXP mkChar(const char* s) {
  const size_t n = strlen(s);
  SEXP x = allocVector(CHARSXP, (R_xlen_t)n);
  memcpy(x->data, s, n + 1);
  return x;
}
SEXP mkString(const char* s) {
  SEXP x = allocVector(STRSXP, 1);
  ((SEXP*)x->data)[0] = mkChar(s);
  return x;
}
SEXP ScalarLogical(int v) { SEXP x = allocVector(LGLSXP, 1); ((int*)x->data)[0] = v; return x; }
SEXP ScalarInteger(int v) { SEXP x = allocVector(INTSXP, 1); ((int*)x->data)[0] = v; return x; }
SEXP ScalarReal(double v) { SEXP x = allocVector(REALSXP, 1); ((double*)x->data)[0] = v; return x; }
char* R_alloc(size_t n, int size) {
  if (size < 0) die("R_alloc: negative size");
  const size_t bytes = n * (size_t)size;
  return (char*)keep(&transient, malloc(bytes ? bytes : 1));
}

SEXPTYPE TYPEOF(SEXP x) { return x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->length; }
int LENGTH(SEXP x) { return (int)x->length; }
double* REAL(SEXP x) { if (x->type != REALSXP) die("REAL() of a non-double"); return (double*)x->data; }
int* INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) die("INTEGER() of a non-integer"); return (int*)x->data; }
int* LOGICAL(SEXP x) { if (x->type != LGLSXP) die("LOGICAL() of a non-logical"); return (int*)x->data; }
const char* CHAR(SEXP x) { if (x->type != CHARSXP) die("CHAR() of a non-CHARSXP"); return (const char*)x->data; }
static void chk(SEXP x, SEXPTYPE t, R_xlen_t i, const char* who) {
  if (x->type != t) die(who);
  if (i < 0 || i >= x->length) die("element index out of range");
}
SEXP STRING_ELT(SEXP x, R_xlen_t i) { chk(x, STRSXP, i, "STRING_ELT of a non-character vector"); return ((SEXP*)x->data)[i]; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { chk(x, VECSXP, i, "VECTOR_ELT of a non-list"); return ((SEXP*)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { chk(x, VECSXP, i, "SET_VECTOR_ELT of a non-list"); ((SEXP*)x->data)[i] = v; return v; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) {
  chk(x, STRSXP, i, "SET_STRING_ELT of a non-character vector");
  if (v->type != CHARSXP) die("SET_STRING_ELT: value is not a CHARSXP");
  ((SEXP*)x->data)[i] = v;
}
SEXP getAttrib(SEXP x, SEXP name) {
  if (x == R_NilValue) return R_NilValue;
  if (name == R_NamesSymbol) return x->names ? x->names : R_NilValue;
  if (name == R_DimSymbol) return x->dim ? x->dim : R_NilValue;
  return R_NilValue;
}
SEXP setAttrib(SEXP x, SEXP name, SEXP val) {
  if (name == R_NamesSymbol) {
    if (val->type != STRSXP || val->length != x->length) die("setAttrib(names): not a character vector of the object's length");
    x->names = val;
  } else if (name == R_DimSymbol) {
    x->dim = val;
  } else {
    die("setAttrib: attribute not modelled by the stub");
  }
  return val;
}

SEXP rapi_protect(SEXP x) { protect_depth++; return x; }
void rapi_unprotect(int n) {
  if (n < 0 || n > protect_depth) die("UNPROTECT of more than is protected");
  protect_depth -= n;
}

void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(rapi_last_error, sizeof rapi_last_error, fmt, ap);
  va_end(ap);
  if (!top) { fprintf(stderr, "Error (no rapi_try frame): %s\n", rapi_last_error); abort(); }
  longjmp(*top, 1);
}
void Rf_warning(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "Warning: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
}
void R_CheckUserInterrupt(void) { rapi_interrupt_checks++; }

int rapi_try(SEXP (*fn)(void*), void* arg, SEXP* out) {
  jmp_buf frame;
  jmp_buf* const outer = top;
  const int depth0 = protect_depth;
  rapi_last_error[0] = 0;
  top = &frame;
  int failed = 0;
  if (setjmp(frame) == 0) {
    SEXP r = fn(arg);
    if (out) *out = r;
    if (protect_depth != depth0) die("protect stack not balanced at the end of a .Call");
  } else {
    failed = 1;
    protect_depth = depth0;           /* R unwinds the protect stack to the context's level on error */
  }
  top = outer;
  free_list(&transient);              /* R_alloc memory lives until the end of the .Call */
  return failed;
}

SEXP rapi_real(const double* v, R_xlen_t n) { SEXP x = allocVector(REALSXP, n); if (n) memcpy(x->data, v, sizeof(double) * (size_t)n); return x; }
SEXP rapi_int(const int* v, R_xlen_t n) { SEXP x = allocVector(INTSXP, n); if (n) memcpy(x->data, v, sizeof(int) * (size_t)n); return x; }
SEXP rapi_lgl(const int* v, R_xlen_t n) { SEXP x = allocVector(LGLSXP, n); if (n) memcpy(x->data, v, sizeof(int) * (size_t)n); return x; }
SEXP rapi_list(const char* name, ...) {
  const char* names[64];
  SEXP vals[64];
  int n = 0;
  va_list ap;
  va_start(ap, name);
  while (name) {
    if (n == 64) die("rapi_list: too many elements");
    names[n] = name;
    vals[n] = va_arg(ap, SEXP);
    n++;
    name = va_arg(ap, const char*);
  }
  va_end(ap);
  SEXP l = allocVector(VECSXP, n), nm = allocVector(STRSXP, n);
  for (int i = 0; i < n; i++) { SET_VECTOR_ELT(l, i, vals[i]); SET_STRING_ELT(nm, i, mkChar(names[i])); }
  setAttrib(l, R_NamesSymbol, nm);
  return l;
}
SEXP rapi_get(SEXP list, const char* name) {
  SEXP nm = getAttrib(list, R_NamesSymbol);
  if (nm == R_NilValue) die("rapi_get: unnamed list");
  for (R_xlen_t i = 0; i < XLENGTH(list); i++)
    if (!strcmp(CHAR(STRING_ELT(nm, i)), name)) return VECTOR_ELT(list, i);
  fprintf(stderr, "rapi_get: no element -%s-\n", name);
  abort();
}
void rapi_free_all(void) { free_list(&all_recs); free_list(&transient); protect_depth = 0; }

int R_registerRoutines(DllInfo* info, const R_CMethodDef* c, const R_CallMethodDef* call, const R_FortranMethodDef* f,
                       const R_ExternalMethodDef* ext) {
  (void)c; (void)f; (void)ext;
  info->call = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) {
  const Rboolean old = info->dynamic_symbols;
  info->dynamic_symbols = value;
  return old;
}
DL_FUNC rapi_lookup(const DllInfo* info, const char* name, int* nargs) {
  for (const R_CallMethodDef* m = info->call; m && m->name; m++)
    if (!strcmp(m->name, name)) { if (nargs) *nargs = m->numArgs; return m->fun; }
  return NULL;
}
