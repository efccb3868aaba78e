/* tests/rapi_stub/Rinternals.h -- a minimal stand-in for the part of R's C API that shim/fmcmc_amd_shim.c uses, so that the
 * shim can be COMPILED (-Wall -Werror, ASan / UBSan) and EXECUTED in an image without R.  TEST INFRASTRUCTURE: written from
 * the API's documented behaviour ("Writing R Extensions", sections 5.9 "Handling R objects in C" and 5.4 "Registering native
 * routines"), not from R's sources; it implements only what the shim calls and checks what R would not (protect-stack
 * balance, out-of-range element access).  A real build uses R's own headers (shim/Makevars).
 *
 * Model: an SEXP points to a heap record {type, length, attrib list (names, dim), payload}; allocVector zero-fills nothing
 * (like R), character vectors hold CHARSXP records, lists hold SEXPs.  error() formats the message into
 * rapi_last_error and longjmps to the harness' frame (rapi_try), as Rf_error unwinds to R's top level.  PROTECT /
 * UNPROTECT keep a counter that rapi_try checks on a normal return. */
#ifndef RAPI_STUB_RINTERNALS_H
#define RAPI_STUB_RINTERNALS_H
#include <stddef.h>
#include <stdint.h>
#include <math.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

typedef struct rapi_sexprec {
  SEXPTYPE type;
  R_xlen_t length;
  struct rapi_sexprec* names;   /* attribute "names" (STRSXP) or NULL */
  struct rapi_sexprec* dim;     /* attribute "dim" (INTSXP) or NULL */
  void* data;                   /* double[] / int[] / char[] / SEXP[] */
} * SEXP;

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;
extern SEXP R_DimSymbol;
extern double R_PosInf, R_NegInf, R_NaReal;
#define NA_REAL R_NaReal
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
typedef int Rboolean;

/* accessors (functions, so that a wrong type or index aborts the test instead of corrupting memory) */
SEXPTYPE TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int LENGTH(SEXP x);
double* REAL(SEXP x);
int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
const char* CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP getAttrib(SEXP x, SEXP name);
SEXP setAttrib(SEXP x, SEXP name, SEXP val);

/* allocation */
SEXP allocVector(SEXPTYPE type, R_xlen_t n);
SEXP allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP alloc3DArray(SEXPTYPE type, int nrow, int ncol, int nface);
SEXP mkChar(const char* s);
SEXP mkString(const char* s);
SEXP ScalarLogical(int v);
SEXP ScalarInteger(int v);
SEXP ScalarReal(double v);
char* R_alloc(size_t n, int size);        /* transient storage, reclaimed at the end of the .Call (here: rapi_try) */

/* protection */
SEXP rapi_protect(SEXP x);
void rapi_unprotect(int n);
#define PROTECT(x) rapi_protect(x)
#define UNPROTECT(n) rapi_unprotect(n)

/* errors */
void Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#define error Rf_error
#define warning Rf_warning

/* ---- harness side (not part of R's API) -------------------------------------------------------------------------- */
extern char rapi_last_error[4096];
extern int rapi_interrupt_checks;          /* calls of R_CheckUserInterrupt so far */
/* Runs fn(arg) as R would run a .Call: returns 0 and *out = the result, or 1 when the call ended in error() (message in
 * rapi_last_error).  Checks that the protect stack is back where it started after a normal return, frees R_alloc memory. */
int rapi_try(SEXP (*fn)(void*), void* arg, SEXP* out);
/* named list from (name, value) pairs, NULL-name terminated; values may be R_NilValue */
SEXP rapi_list(const char* name, ...);
SEXP rapi_real(const double* v, R_xlen_t n);
SEXP rapi_int(const int* v, R_xlen_t n);
SEXP rapi_lgl(const int* v, R_xlen_t n);
SEXP rapi_get(SEXP list, const char* name);   /* element by name, aborts when absent */
void rapi_free_all(void);                     /* releases every record allocated so far (end of a test) */

#ifdef __cplusplus
}
#endif
#endif
